// invres.hip -- a whole InvertedResidual block (models/mobilenetv2.py:39-63) as ONE kernel for the small-map stages of
// MobileNetV2 (28 x 28, 14 x 14, 7 x 7):
//     1x1 expand (cin -> hidden) + norm + ReLU6  ->  3x3 depthwise (stride 1 | 2) + norm + ReLU6  ->  1x1 project (hidden -> cout)
//     + norm  [-> + x]
// As three launches (convnorm.hip) the 6x-wide hidden tensor makes two HBM / L2 round trips and every launch is 256-512
// short-lived workgroups: the pointwise layers of these stages ran at 15 % of the HBM roof AND 15 % of the MFMA roof
// (profiles/r02_perf_mobilenet_v2_b64_layers.log).  Here the hidden tensor never leaves the CU:
//
//   workgroup   = (a region of output pixels: a 7-row strip of a 28-wide map, a whole 14 x 14 map, or two 7 x 7 images)
//                 x (a SLICE of the hidden channels); the region's input pixels x ALL cin channels are staged in LDS once
//   per chunk of 32 hidden channels of its slice:
//     phase 1   expand on the fp32 MFMA (rows = input pixels, columns = the chunk's channels, K = cin, one ascending-k chain
//               per output) -> norm -> ReLU6 -> LDS tile hid[channel][pixel]
//     phase 2   3x3 depthwise on that tile (VALU, the oracle's 9-tap fmaf chain from +0 in (ky, kx) order, zero padding fed
//               through the chain) -> norm -> ReLU6 -> LDS tile dws[channel][output pixel]
//     phase 3   project partial on the MFMA: acc[output pixel][cout] += sum over the chunk's 32 channels, ascending; the
//               accumulators stay in registers across the slice's chunks
//   epilogue    one slice: norm (+ x) and store.  Several slices (launches that would otherwise leave most CUs idle: batch 64
//               is 64 regions on 256 CUs): each workgroup writes its raw partial sums to the caller's workspace and
//               k_invres_reduce adds the slices IN ASCENDING ORDER, then norm (+ x).
//
// Summation order (part of the ABI, stated by mv_inverted_residual_k_slices and restated by the oracle as the composition
// orc_conv2d_affine_act_f32 (expand: single chain) -> orc_conv2d_affine_act_f32 (depthwise) ->
// orc_pointwise_sliced_affine_act_f32 (project: one chain per slice from +0, slices added in ascending order)): bit-exact.
#include <cstdlib>
#include <type_traits>

#include "mv_common.h"
#include "mv_act.h"
#include "mv_invres.h"

namespace mv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-byte global access (gfx950 takes it)

constexpr int kHC = 32;        // hidden channels per chunk = one MFMA column tile


struct IrArgs {
  const float* x;                 // (n, cin, H, W)
  const float* w1;                // (hidden, cin)
  const float *a1, *b1;           // folded norm after the expansion: [hidden]
  const float* wd;                // (hidden, 3, 3)
  const float *a2, *b2;           // after the depthwise conv: [hidden]
  const float* w2;                // (cout, hidden)
  const float *a3, *b3;           // after the projection: [cout]
  const float* res;               // x when the block adds its input (stride 1, cin == cout), else null
  float* y;                       // (n, cout, OH, OW)
  float* part;                    // [slices][n][cout][OH*OW] raw partial sums (slices > 1)
  int n, hidden, cout, affine;
  int imgs, strips;               // images per region (7 x 7 maps), strips per image
  int slices, cps;                // K slices of the projection = hidden slices across workgroups; chunks per slice
  int hp;                         // LDS pitch (floats) of the hidden tile
  int off_hid, off_dws, off_w2, off_terms;  // LDS offsets (floats); w1s starts at 0
};

__device__ __forceinline__ float ir_norm(float v, float a, float b, int affine) {
  float two = v * a;  // FrozenBatchNorm2d: x * scale, then + bias (two roundings)
  two = two + b;
  const float one = fmaf(v, a, b);  // BatchNorm2d (eval)
  return affine == 2 ? one : two;
}
// the same with the (launch-uniform) mode as a type: the element loops of the kernel are instantiated per mode and entered through
// ONE branch, instead of computing both forms and selecting per element
template <bool FMA>
__device__ __forceinline__ float ir_norm_t(float v, float a, float b) {
  if (FMA) return fmaf(v, a, b);
  const float t = v * a;
  return t + b;
}
__device__ __forceinline__ float ir_relu6(float v) {  // NaN passes through, like `v < 0 ? 0 : v`
  return clamp_f32(v, 0.f, 6.f);
}

constexpr int kOP = 36;  // pitch of the [row][k parity][k / 2] operand tiles of one 32-channel chunk: 16-byte reads of consecutive
                         // rows fall on distinct banks (pitch / 4 odd)
constexpr int kTB = 36;  // pitch of a wave's [channel][pixel] transpose buffer

// W: map side (7, 14, 28; square maps).  PTOUT: 32-pixel tiles of the region's OUTPUT pixels.  COT = cout / 32.  CINQ = cin / 32.
//
// Operand layouts.  v_mfma_f32_32x32x2_f32 takes, per lane, A[row = lane % 32][k = lane / 32] and B[k = lane / 32][col = lane % 32];
// a k-step covers k = 2 s, 2 s + 1.  A lane therefore walks ONE parity of k: tiles stored as [row][parity][s] give it four
// consecutive k-steps per 16-byte LDS read instead of one 4-byte read per step -- at one wave per SIMD every LDS instruction
// costs the MFMA stream ~20 cycles (profiles/r03_trace_invres_v1.log: 218 cycles per k-step of two MFMAs with three 4-byte reads).
//   expansion   A = the region's input pixels: the SAME for every chunk -> loaded once from global memory into registers
//               (areg[tile][s], <= 96 VGPRs); B = w1s[channel][parity][s] from LDS
//   projection  A = dws[output pixel][parity][s] (written in that layout by the depthwise phase), B = w2s[cout][parity][s]
// waves split of the projection's PTOUT x COT output tiles: (PG pixel groups) x (CG = NW / PG channel groups) with the fewest
// tiles per wave
constexpr int ir_tiles_per_wave(int ptout, int cot, int pg, int nw) { return ((ptout + pg - 1) / pg) * ((cot + nw / pg - 1) / (nw / pg)); }
constexpr int ir_pick_pg(int ptout, int cot, int nw) {
  int best = 1;
  for (int pg = 1; pg <= nw; pg *= 2)
    if (ir_tiles_per_wave(ptout, cot, pg, nw) < ir_tiles_per_wave(ptout, cot, best, nw)) best = pg;
  return best;
}

// NW = waves per workgroup: 4, or 8 (two per SIMD: every phase's LDS / MFMA latencies overlap across the pair; the wide kernel's
// sweep measured 8 waves 15-20 % ahead of 4 at the same region)
template <int W, int STRIDE, int PTOUT, int COT, int CINQ, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_invres(const IrArgs A) {
  constexpr int NTHR = 64 * NW;
  constexpr int H = W, OW = (W - 1) / STRIDE + 1, OH = OW;
  constexpr int CIN = 32 * CINQ, COUT = 32 * COT, KS1 = CIN / 2;
  constexpr int ORH = (W == 14 && STRIDE == 1) ? 14 : 7;    // output rows per region
  constexpr int OPI = ORH * OW;                             // output pixels per image of a region
  constexpr int W1P = (CIN / 4) % 2 ? CIN : CIN + 4;        // pitch of w1s: [channel][parity][KS1], pitch / 4 odd
  constexpr int NPINMAX = (W == 28) ? ((ORH - 1) * STRIDE + 3) * W : (W == 14 ? 196 : 98);  // the region's input pixels, at most
  constexpr int T1 = (NPINMAX + 31) / 32, NT1 = (T1 + NW - 1) / NW;  // expansion tiles of a region / per wave
  constexpr int PG = ir_pick_pg(PTOUT, COT, NW);  // the waves as PG pixel groups x CG channel groups (projection)
  constexpr int CG = NW / PG;
  constexpr int W1Q = (8 * CIN + NTHR - 1) / NTHR, W2Q = (8 * COUT + NTHR - 1) / NTHR;  // 16-byte pieces of the chunk's weights per thread
  constexpr int PTW = (PTOUT + PG - 1) / PG, CTW = (COT + CG - 1) / CG;  // output tiles a wave owns
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const w1s = lds;                // [32][W1P]          expand weights of the chunk
  float* const hid = lds + A.off_hid;    // [32][hp]           expanded + norm + ReLU6 (also the epilogue's transpose buffers)
  float* const dws = lds + A.off_dws;    // [32 * PTOUT][36]   depthwise + norm + ReLU6, [output pixel][parity][s]
  float* const w2s = lds + A.off_w2;     // [COUT][36]         project weights of the chunk, [cout][parity][s]
  float* const t1a = lds + A.off_terms;  // [32] x 4 norm terms, then [32][9] depthwise taps
  float* const t1b = t1a + 32;
  float* const t2a = t1a + 64;
  float* const t2b = t1a + 96;
  float* const wds = t1a + 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hf = lane >> 5;
#ifdef MV_IR_TRACE
  // tools/trace_invres.py: cycle stamps of workgroup (0, 0)'s four waves, written behind y (the caller over-allocates)
  int trace_slot = 0;
  long long* const trace_buf = reinterpret_cast<long long*>(A.y + (size_t)A.n * COUT * OH * OW) + wave * 64;
#define MV_IR_STAMP() do { if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && trace_slot < 64) trace_buf[trace_slot++] = clock64(); } while (0)
#else
#define MV_IR_STAMP() do { } while (0)
#endif
  MV_IR_STAMP();  // 0: start
  const int strip = blockIdx.x % A.strips, ig = blockIdx.x / A.strips;
  const int slice = blockIdx.y;
  const int img0 = ig * A.imgs;
  const int imgs = min(A.imgs, A.n - img0);
  const int oy0 = strip * ORH;
  const int iy_lo = max(oy0 * STRIDE - 1, 0), iy_hi = min((oy0 + ORH - 1) * STRIDE + 1, H - 1);
  const int RH = iy_hi - iy_lo + 1;                      // input rows of the region
  const int npin = imgs * RH * W, npout = imgs * OPI;
  const int HP = A.hp;
  const int chunks = A.hidden / kHC;
  const int ch0 = slice * A.cps, ch1 = min(ch0 + A.cps, chunks);

  // ---- chunk operands: global -> registers (in flight during the previous chunk's phases) -> LDS
  f32x4 w1r[W1Q], w2r[W2Q];
  float tr = 0.f, wdr[2] = {0.f, 0.f};
  auto gload = [&](int ch) {
    const int h0 = ch * kHC;
#pragma unroll
    for (int u = 0; u < W1Q; ++u) {  // 32 rows x CIN floats, contiguous (threads past the end repeat the last piece and store nothing)
      const int idx = min(tid + NTHR * u, 8 * CIN - 1);
      w1r[u] = *reinterpret_cast<const f32x4*>(A.w1 + (size_t)h0 * CIN + 4 * idx);
    }
#pragma unroll
    for (int u = 0; u < W2Q; ++u) {   // COUT rows x 32 floats at column h0
      const int idx = min(tid + NTHR * u, 8 * COUT - 1);
      const int row = idx >> 3, q = idx & 7;
      w2r[u] = *reinterpret_cast<const f32x4*>(A.w2 + (size_t)row * A.hidden + h0 + 4 * q);
    }
    {
      const float* src = tid < 32 ? A.a1 : (tid < 64 ? A.b1 : (tid < 96 ? A.a2 : A.b2));
      tr = src[h0 + (tid & 31)];  // threads from 128 on load a value nobody stores: every load stays unconditional
    }
    wdr[0] = A.wd[(size_t)h0 * 9 + (tid & 255)];
    wdr[1] = A.wd[(size_t)h0 * 9 + 256 + (tid & 31)];
  };
  auto lstore = [&]() {  // k = 4 q .. 4 q + 3 of a row -> parities 0, 1, 0, 1 at s = 2 q, 2 q, 2 q + 1, 2 q + 1
#pragma unroll
    for (int u = 0; u < W1Q; ++u) {
      const int idx = tid + NTHR * u;
      if (8 * CIN % NTHR == 0 || idx < 8 * CIN) {
        const int row = idx / (CIN / 4), q = idx % (CIN / 4);
        float* d = w1s + row * W1P + 2 * q;
        *reinterpret_cast<f32x2*>(d) = (f32x2){w1r[u].x, w1r[u].z};
        *reinterpret_cast<f32x2*>(d + KS1) = (f32x2){w1r[u].y, w1r[u].w};
      }
    }
#pragma unroll
    for (int u = 0; u < W2Q; ++u) {
      const int idx = tid + NTHR * u;
      if (8 * COUT % NTHR == 0 || idx < 8 * COUT) {
        const int row = idx >> 3, q = idx & 7;
        float* d = w2s + row * kOP + 2 * q;
        *reinterpret_cast<f32x2*>(d) = (f32x2){w2r[u].x, w2r[u].z};
        *reinterpret_cast<f32x2*>(d + 16) = (f32x2){w2r[u].y, w2r[u].w};
      }
    }
    if (tid < 128) t1a[tid] = tr;
    if (tid < 256) wds[tid] = wdr[0];
    if (tid < 32) wds[256 + tid] = wdr[1];
  };
  if (ch0 < ch1) gload(ch0);
  if (tid < 32) wds[288 + tid] = 0.f;  // the zero row (visible after the first chunk's barriers)

  // ---- the expansion's A operand: this wave's input pixel tiles x all CIN channels, from global memory into registers, once.
  //      Lane (l31, hf) of tile t holds x[k = 2 s + hf][pixel (wave + 4 t) * 32 + l31] for every k-step s; 32 lanes read 128
  //      contiguous bytes of a channel row.  Pixels past the region read a clamped address (rows of the tile nobody stores).
  float areg[NT1][KS1];
  {
#pragma unroll
    for (int t = 0; t < NT1; ++t) {
      const int p = min((wave + NW * t) * 32 + l31, npin - 1);
      size_t base;
      if ((W & 1) == 0) {
        base = (size_t)img0 * CIN * (H * W) + iy_lo * W + p;   // one image per region: its pixels are contiguous
      } else {
        const int im = p >= RH * W ? 1 : 0;                     // 7 x 7: up to two whole images
        base = (size_t)(img0 + im) * CIN * (H * W) + (p - im * (RH * W));
      }
      const float* src = A.x + base + (size_t)hf * (H * W);
#pragma unroll
      for (int s = 0; s < KS1; ++s) areg[t][s] = src[(size_t)(2 * s) * (H * W)];
    }
  }

  // ---- projection accumulators: wave (pg, cg) owns output pixel tiles pg + PG * i and channel tiles cg + CG * j
  const int pg = wave % PG, cg = wave / PG;
  f32x16 acc[PTW][CTW];
#pragma unroll
  for (int i = 0; i < PTW; ++i)
#pragma unroll
    for (int j = 0; j < CTW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  MV_IR_STAMP();  // 1: loads issued
  for (int ch = ch0; ch < ch1; ++ch) {
    __syncthreads();  // the previous chunk's projection has read w2s / dws
    lstore();
    __syncthreads();
    MV_IR_STAMP();  // per chunk +0: operands in LDS
    if (ch + 1 < ch1) gload(ch + 1);

    // ---- phase 1: hid[c][p] = ReLU6(norm(sum_k x[k][p] * w1[c][k])): NT1 tiles per wave at once, B from LDS 4 k-steps per read
    {
      const float na = t1a[l31], nb = t1b[l31];
      f32x16 c[NT1];
#pragma unroll
      for (int t = 0; t < NT1; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[t][r] = 0.f;
      const float* bp = w1s + l31 * W1P + hf * KS1;
      f32x4 bq[2];
      bq[0] = *reinterpret_cast<const f32x4*>(bp);
#pragma unroll
      for (int j = 0; j < KS1 / 4; ++j) {
        if (j + 1 < KS1 / 4) bq[(j + 1) & 1] = *reinterpret_cast<const f32x4*>(bp + 4 * (j + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < NT1; ++t)
            c[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[t][4 * j + i], bq[j & 1][i], c[t], 0, 0, 0);  // rows = pixels, columns = channels
        __builtin_amdgcn_sched_barrier(0);
      }
      // the tile's rows up to 32 * (tiles of the region) fit the row pitch: 16-byte stores, no per-pixel bounds
      float* hrow = hid + l31 * HP;
      auto store_tiles = [&](auto fma_mode) {
        constexpr bool FMA = decltype(fma_mode)::value;
#pragma unroll
        for (int t = 0; t < NT1; ++t) {
          const int pt = wave + NW * t;
          if (pt * 32 >= npin) continue;  // wave-uniform: a tile past the region
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ir_relu6(ir_norm_t<FMA>(c[t][4 * g + j], na, nb));
            *reinterpret_cast<f32x4*>(hrow + pt * 32 + 8 * g + 4 * hf) = v;
          }
        }
      };
      if (A.affine == 2) store_tiles(std::true_type{}); else store_tiles(std::false_type{});
    }
    MV_IR_STAMP();  // +1: expansion done
    __syncthreads();
    MV_IR_STAMP();  // +2

    // ---- phase 2: 3x3 depthwise of the chunk's 32 channels; a thread computes one output row from three tile rows in registers
    //      and writes it as the projection's A operand: dws[output pixel][channel parity][channel / 2].  A row above / below the
    //      image is read from a row of zeros kept behind the terms (no per-element select).
    auto depthwise = [&](auto fma_mode) {
      constexpr bool FMA = decltype(fma_mode)::value;
      const int items = imgs * kHC * ORH;
      const float* const zrow = wds + 288;
      for (int it = tid; it < items; it += NTHR) {
        const int oyl = it % ORH, t2 = it / ORH;
        const int c = t2 & (kHC - 1), im = t2 >> 5;
        const int oy = oy0 + oyl;
        float wk[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) wk[i] = wds[c * 9 + i];
        const float na = t2a[c], nb = t2b[c];
        float rows[3][W + 2];  // columns -1 .. W
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = oy * STRIDE - 1 + ky;
          const float* rp = (iy >= 0 && iy < H) ? hid + c * HP + (im * RH + iy - iy_lo) * W : zrow;
          rows[ky][0] = 0.f, rows[ky][W + 1] = 0.f;
          if ((W & 3) == 0) {
#pragma unroll
            for (int i = 0; i < W / 4; ++i) {
              const f32x4 q = *reinterpret_cast<const f32x4*>(rp + 4 * i);
              rows[ky][4 * i + 1] = q.x, rows[ky][4 * i + 2] = q.y, rows[ky][4 * i + 3] = q.z, rows[ky][4 * i + 4] = q.w;
            }
          } else if ((W & 1) == 0) {
#pragma unroll
            for (int i = 0; i < W / 2; ++i) {
              const f32x2 q = *reinterpret_cast<const f32x2*>(rp + 2 * i);
              rows[ky][2 * i + 1] = q.x, rows[ky][2 * i + 2] = q.y;
            }
          } else {
#pragma unroll
            for (int i = 0; i < W; ++i) rows[ky][i + 1] = rp[i];
          }
        }
        float* op = dws + (im * OPI + oyl * OW) * kOP + (c & 1) * 16 + (c >> 1);
#pragma unroll
        for (int ox = 0; ox < OW; ++ox) {
          float a = fmaf(wk[0], rows[0][ox * STRIDE], 0.f);
          a = fmaf(wk[1], rows[0][ox * STRIDE + 1], a);
          a = fmaf(wk[2], rows[0][ox * STRIDE + 2], a);
          a = fmaf(wk[3], rows[1][ox * STRIDE], a);
          a = fmaf(wk[4], rows[1][ox * STRIDE + 1], a);
          a = fmaf(wk[5], rows[1][ox * STRIDE + 2], a);
          a = fmaf(wk[6], rows[2][ox * STRIDE], a);
          a = fmaf(wk[7], rows[2][ox * STRIDE + 1], a);
          a = fmaf(wk[8], rows[2][ox * STRIDE + 2], a);
          op[ox * kOP] = ir_relu6(ir_norm_t<FMA>(a, na, nb));
        }
      }
    };
    if (A.affine == 2) depthwise(std::true_type{}); else depthwise(std::false_type{});
    MV_IR_STAMP();  // +3: depthwise done
    __syncthreads();
    MV_IR_STAMP();  // +4

    // ---- phase 3: acc[q][co] += sum over the chunk's channels (ascending) dws[q][c] * w2[co][c]: both operands 4 k-steps per
    //      16-byte read; no branch inside the loop (a tile past the region / past cout multiplies rows nobody stores)
    {
      const float* ap = dws + (pg * 32 + l31) * kOP + hf * 16;
      const float* bp = w2s + (cg * 32 + l31) * kOP + hf * 16;
      int aoff[PTW], boff[CTW];
#pragma unroll
      for (int i = 0; i < PTW; ++i) aoff[i] = min(i * PG, PTOUT - 1 - pg) * (32 * kOP);
#pragma unroll
      for (int j = 0; j < CTW; ++j) boff[j] = min(j * CG, COT - 1 - cg) * (32 * kOP);
      f32x4 aq[2][PTW], bq[2][CTW];
#pragma unroll
      for (int i = 0; i < PTW; ++i) aq[0][i] = *reinterpret_cast<const f32x4*>(ap + aoff[i]);
#pragma unroll
      for (int j = 0; j < CTW; ++j) bq[0][j] = *reinterpret_cast<const f32x4*>(bp + boff[j]);
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // 4 groups of 4 k-steps
        if (g + 1 < 4) {
#pragma unroll
          for (int i = 0; i < PTW; ++i) aq[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(ap + aoff[i] + 4 * (g + 1));
#pragma unroll
          for (int j = 0; j < CTW; ++j) bq[(g + 1) & 1][j] = *reinterpret_cast<const f32x4*>(bp + boff[j] + 4 * (g + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < PTW; ++i)
#pragma unroll
            for (int j = 0; j < CTW; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[g & 1][i][e], bq[g & 1][j][e], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    MV_IR_STAMP();  // +5: projection of the chunk done
  }

  // ---- epilogue.  The accumulators hold (pixel rows) x (one channel per lane): stored like that, one instruction would write
  //      16 bytes into each of 64 different cache lines.  Every 32 x 32 tile goes through a wave-private LDS buffer
  //      [channel][pixel] (the hidden tile is dead now) and comes back with 8 consecutive lanes covering one channel's 32
  //      pixels: one store instruction writes 8 full 128-byte lines.
  __syncthreads();
  float* const tb = hid + wave * (32 * kTB);
  const size_t plane = (size_t)OH * OW;
  const bool final_pass = A.slices == 1;
  float* const dst = final_pass ? A.y : A.part + (size_t)slice * A.n * COUT * plane;
  const int r0 = lane >> 3, q4 = (lane & 7) * 4;  // after the transpose: channel rows r0 + 8 j, pixels q4 .. q4 + 3
#pragma unroll
  for (int j = 0; j < CTW; ++j) {
    const int ct = cg + CG * j;
    if (ct >= COT) continue;
#pragma unroll
    for (int i = 0; i < PTW; ++i) {
      const int pt = pg + PG * i;
      if (pt >= PTOUT || pt * 32 >= npout) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(tb + l31 * kTB + 8 * g + 4 * hf) =
            (f32x4){acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
      const int q = pt * 32 + q4;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int co = ct * 32 + r0 + 8 * jj;
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(tb + (r0 + 8 * jj) * kTB + q4);
        if (q >= npout) continue;
        float v[4] = {t4.x, t4.y, t4.z, t4.w};
        if (final_pass) {
          const float na = A.a3[co], nb = A.b3[co];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ir_norm(v[r], na, nb, A.affine);
        }
        const int im = q / OPI, qi = q - im * OPI;
        const size_t o = ((size_t)(img0 + im) * COUT + co) * plane + (size_t)oy0 * OW + qi;
        if (q + 3 < npout && qi + 3 < OPI) {  // the 4 pixels lie in one image: one (dword-aligned) 16-byte access
          if (final_pass && A.res) {
            const f32x4u rv = *reinterpret_cast<const f32x4u*>(A.res + o);
            v[0] = rv.x + v[0], v[1] = rv.y + v[1], v[2] = rv.z + v[2], v[3] = rv.w + v[3];
          }
          *reinterpret_cast<f32x4u*>(dst + o) = (f32x4u){v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (q + r >= npout) break;
            const int im2 = (q + r) / OPI, q2 = (q + r) - im2 * OPI;
            const size_t o2 = ((size_t)(img0 + im2) * COUT + co) * plane + (size_t)oy0 * OW + q2;
            float e = v[r];
            if (final_pass && A.res) e = A.res[o2] + e;
            dst[o2] = e;
          }
        }
      }
    }
  }
  MV_IR_STAMP();  // last: outputs stored
}


// ================================================================================================ wide maps: 112 x 112 and 56 x 56
// MobileNetV2's first three expanding blocks (16 -> 96 -> 24 on 112 x 112 at stride 2; 24 -> 144 -> 24 on 56 x 56; 24 -> 144 -> 32
// at stride 2) move the widest hidden tensors of the net -- 308 MB written and read back at batch 64 for the first one alone --
// and ran as three HBM-bound launches at 45-50 % of the roof (profiles/r03_ktrace_mobilenet_v2_b64.log: 234 + 139 + 92 us).
// The same three phases as k_invres, with
//   region     ORH output rows of ONE image at full width: (ORH - 1) * STRIDE + 3 input rows are expanded per region, the halo
//              rows twice (the 1x1 expansion has K = 16 / 24: recomputing it costs a few MFMAs, storing it costs HBM)
//   channels   cin in {16, 24}, hidden and cout need not fill their last 32-wide tile: rows / columns past them are staged as
//              zeros -- a zero hidden channel adds +0 products to the projection's chain, an exact no-op from +0
//   depthwise  a thread computes 8 consecutive outputs of one row (rows of 56 / 112 pixels do not fit a thread's registers):
//              lanes 0-31 walk the chunk's 32 channels, so the 16-byte reads of hid[channel][pixel] (pitch / 4 odd) and the
//              4-byte stores into dws[pixel][parity][channel / 2] are both conflict-free
template <int W, int STRIDE, int ORH, int CIN, int NW, int OCC, bool EXPAND>
__global__ __launch_bounds__(64 * NW, OCC) void k_invres_wide(const IrArgs A) {
  constexpr int NTHR = 64 * NW;
  constexpr int H = W, OW = W / STRIDE;
  constexpr int KS1 = CIN / 2;
  constexpr int RHMAX = (ORH - 1) * STRIDE + 3;
  constexpr int T1 = (RHMAX * W + 31) / 32, NT1 = (T1 + NW - 1) / NW;   // expansion tiles of a region / per wave
  constexpr int OPI = ORH * OW, PTOUT = (OPI + 31) / 32;
  constexpr int W1P = (CIN / 4) % 2 ? CIN : CIN + 4;
  constexpr int W1Q = EXPAND ? (8 * CIN + NTHR - 1) / NTHR : 0;
  constexpr int P4 = RHMAX * W / 4, NLOAD = EXPAND ? 0 : (CIN * P4 + NTHR - 1) / NTHR;  // !EXPAND: 16-byte pieces of a region's input per thread
                     // 16-byte pieces of a chunk's expand weights per thread
  constexpr int PG = PTOUT < NW ? PTOUT : NW;                     // cout <= 32: one channel tile; waves >= PG sit the projection out
  constexpr int PTW = (PTOUT + PG - 1) / PG;
  constexpr int NSEG = (OW + 7) / 8;
  static_assert(OW % 4 == 0 && (OW * OW) % 4 == 0 && OW % ORH == 0 && KS1 % 4 == 0, "geometry");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const w1s = lds;
  float* const hid = lds + A.off_hid;
  float* const dws = lds + A.off_dws;
  float* const w2s = lds + A.off_w2;
  float* const t1a = lds + A.off_terms;
  float* const t1b = t1a + 32;
  float* const t2a = t1a + 64;
  float* const t2b = t1a + 96;
  float* const wds = t1a + 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hf = lane >> 5;
#ifdef MV_IR_TRACE
  int trace_slot = 0;
  long long* const trace_buf = reinterpret_cast<long long*>(A.y + (size_t)A.n * A.cout * OW * OW) + wave * 64;
#endif
  MV_IR_STAMP();  // 0: start
  const int slice = blockIdx.y;
  const int nregions = A.n * A.strips;
  const int HP = A.hp;
  const int chunks = (A.hidden + kHC - 1) / kHC;
  const int ch0 = slice * A.cps, ch1 = min(ch0 + A.cps, chunks);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  f32x4 w1r[W1Q > 0 ? W1Q : 1], w2r;
  float tr = 0.f, wdr[2] = {0.f, 0.f};
  // Loads are issued one chunk (one region) ahead and must stay in flight: every address is clamped into the tensor and NOTHING
  // looks at the loaded value here -- a `valid ? v : 0` next to the load is a use, the compiler waits for the data right there and
  // the prefetch becomes a blocking load (the first version: 3 us at every region's start).  Rows / columns past hidden / cout
  // become zeros in lstore(), which recomputes the same predicates.
  auto gload = [&](int ch) {
    const int h0 = ch * kHC;
#pragma unroll
    for (int u = 0; u < W1Q; ++u) {
      const int idx = tid + NTHR * u;
      const bool ok = idx < 8 * CIN && h0 + idx / (CIN / 4) < A.hidden;
      w1r[u] = *reinterpret_cast<const f32x4*>(A.w1 + (ok ? (size_t)h0 * CIN + 4 * idx : 0));
    }
    {
      const int row = (tid >> 3) & 31, q = tid & 7;  // threads past 255 repeat the first 256's loads and store nothing
      const bool ok = row < A.cout && h0 + 4 * q < A.hidden;
      w2r = *reinterpret_cast<const f32x4*>(A.w2 + (ok ? (size_t)row * A.hidden + h0 + 4 * q : 0));
    }
    {
      const float* src = tid < 32 ? (EXPAND ? A.a1 : A.a2) : (tid < 64 ? (EXPAND ? A.b1 : A.b2) : (tid < 96 ? A.a2 : A.b2));
      tr = src[min(h0 + (tid & 31), A.hidden - 1)];
    }
    const int lim = A.hidden * 9 - 1;
    wdr[0] = A.wd[min(h0 * 9 + (tid & 255), lim)];
    wdr[1] = A.wd[min(h0 * 9 + 256 + (tid & 31), lim)];
  };
  auto lstore = [&](int ch) {
    const int h0 = ch * kHC;
#pragma unroll
    for (int u = 0; u < W1Q; ++u) {
      const int idx = tid + NTHR * u;
      if (idx < 8 * CIN) {
        const int row = idx / (CIN / 4), q = idx % (CIN / 4);
        const f32x4 v = h0 + row < A.hidden ? w1r[u] : zero4;
        float* d = w1s + row * W1P + 2 * q;
        *reinterpret_cast<f32x2*>(d) = (f32x2){v.x, v.z};
        *reinterpret_cast<f32x2*>(d + KS1) = (f32x2){v.y, v.w};
      }
    }
    if (tid < 256) {
      const int row = tid >> 3, q = tid & 7;
      const f32x4 v = (row < A.cout && h0 + 4 * q < A.hidden) ? w2r : zero4;
      float* d = w2s + row * kOP + 2 * q;
      *reinterpret_cast<f32x2*>(d) = (f32x2){v.x, v.z};
      *reinterpret_cast<f32x2*>(d + 16) = (f32x2){v.y, v.w};
    }
    const int lim = A.hidden * 9 - 1;
    if (tid < 128) t1a[tid] = h0 + (tid & 31) < A.hidden ? tr : 0.f;
    if (tid < 256) wds[tid] = h0 * 9 + tid <= lim ? wdr[0] : 0.f;
    if (tid < 32) wds[256 + tid] = h0 * 9 + 256 + tid <= lim ? wdr[1] : 0.f;
  };
  if (ch0 < ch1) gload(ch0);
  if (tid < W + 4) wds[288 + tid] = 0.f;  // a row of zeros: the depthwise conv's padding rows

  // the expansion's A operand: the region's input pixels (contiguous in every channel plane) x all CIN channels, in registers.
  // A workgroup walks regions blockIdx.x, + gridDim.x, ...: the NEXT region's pixels are loaded while this one is computed -- at one
  // workgroup per CU (the hidden tile fills the LDS) a region's ~2 us of load latency at its start was a fifth of its time
  // (profiles/r03_trace_invres_wide_v1.log)
  struct Region { int img, oy0, iy_lo, npin; };
  auto region_of = [&](int r) {
    Region g;
    g.img = r / A.strips;
    g.oy0 = (r - g.img * A.strips) * ORH;
    g.iy_lo = max(g.oy0 * STRIDE - 1, 0);
    g.npin = (min((g.oy0 + ORH - 1) * STRIDE + 1, H - 1) - g.iy_lo + 1) * W;
    return g;
  };
  float areg[NT1][KS1], anext[NT1][KS1];
  f32x4 xnext[NLOAD > 0 ? NLOAD : 1];  // !EXPAND (a block without expansion: hidden = cin): the region's raw input, on its way to `hid`
  auto aload = [&](const Region& g) {
    if constexpr (EXPAND) {
#pragma unroll
      for (int t = 0; t < NT1; ++t) {
        const int p = min((wave + NW * t) * 32 + l31, g.npin - 1);
        const float* src = A.x + ((size_t)g.img * CIN + hf) * (H * W) + g.iy_lo * W + p;
#pragma unroll
        for (int s = 0; s < KS1; ++s) anext[t][s] = src[(size_t)(2 * s) * (H * W)];
      }
    } else {
#pragma unroll
      for (int u = 0; u < NLOAD; ++u) {
        const int idx = tid + NTHR * u, c = idx / P4, p4 = idx - c * P4;
        const bool ok = c < CIN && 4 * p4 < g.npin;  // rows past the region are stored as whatever the clamped address holds: nobody reads them
        xnext[u] = *reinterpret_cast<const f32x4*>(A.x + (ok ? ((size_t)g.img * CIN + c) * (H * W) + g.iy_lo * W + 4 * p4 : 0));
      }
    }
  };
  auto xstore = [&]() {
#pragma unroll
    for (int u = 0; u < NLOAD; ++u) {
      const int idx = tid + NTHR * u, c = idx / P4, p4 = idx - c * P4;
      if (c < CIN) *reinterpret_cast<f32x4*>(hid + c * HP + 4 * p4) = xnext[u];
    }
  };
  const int pg = wave % PG, cg = wave / PG;  // cg > 0: no projection tile
  if ((int)blockIdx.x < nregions) aload(region_of(blockIdx.x));

  for (int region = blockIdx.x; region < nregions; region += gridDim.x) {
  const Region rg = region_of(region);
  const int img = rg.img, oy0 = rg.oy0, iy_lo = rg.iy_lo, npin = rg.npin;
  const bool more = region + (int)gridDim.x < nregions;
  if constexpr (EXPAND) {
#pragma unroll
    for (int t = 0; t < NT1; ++t)
#pragma unroll
      for (int s = 0; s < KS1; ++s) areg[t][s] = anext[t][s];
    if (more) aload(region_of(region + gridDim.x));
  }

  f32x16 acc[PTW];
#pragma unroll
  for (int i = 0; i < PTW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  MV_IR_STAMP();  // 1: loads issued
  for (int ch = ch0; ch < ch1; ++ch) {
    __syncthreads();
    lstore(ch);
    if constexpr (!EXPAND) xstore();  // hidden = the input itself: one chunk, `hid` filled by a flat copy
    __syncthreads();
    MV_IR_STAMP();  // per chunk +0: operands in LDS
    if (ch + 1 < ch1) gload(ch + 1);
    else if (more) gload(ch0);  // the next region starts with this slice's first chunk again
    if constexpr (!EXPAND) {
      if (more) aload(region_of(region + gridDim.x));
    }

    // ---- phase 1: expansion -> norm -> ReLU6 -> hid[channel][region pixel].  The region's T1 tiles rarely divide by the waves: a
    //      wave whose last tile lies past the region runs the loop instantiated for one tile less (wave-uniform choice OUTSIDE the
    //      MFMA loop) instead of multiplying a clamped tile nobody stores -- 18 tiles on 8 waves were 24 tiles' worth of MFMAs
    auto expand = [&](auto ntc, auto fma_mode) {
      constexpr int NT = decltype(ntc)::value;
      constexpr bool FMA = decltype(fma_mode)::value;
      if constexpr (NT > 0) {
        const float na = t1a[l31], nb = t1b[l31];
        f32x16 c[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) c[t][r] = 0.f;
        const float* bp = w1s + l31 * W1P + hf * KS1;
        f32x4 bq[2];
        bq[0] = *reinterpret_cast<const f32x4*>(bp);
#pragma unroll
        for (int j = 0; j < KS1 / 4; ++j) {
          if (j + 1 < KS1 / 4) bq[(j + 1) & 1] = *reinterpret_cast<const f32x4*>(bp + 4 * (j + 1));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t)
              c[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(areg[t][4 * j + i], bq[j & 1][i], c[t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        float* hrow = hid + l31 * HP;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int pt = wave + NW * t;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ir_relu6(ir_norm_t<FMA>(c[t][4 * g + j], na, nb));
            *reinterpret_cast<f32x4*>(hrow + pt * 32 + 8 * g + 4 * hf) = v;
          }
        }
      }
    };
    if constexpr (EXPAND) {
      const bool last_tile = (wave + NW * (NT1 - 1)) * 32 < npin;  // < T1 * 32 by construction of npin
      if (A.affine == 2) {
        if (last_tile) expand(std::integral_constant<int, NT1>{}, std::true_type{});
        else expand(std::integral_constant<int, NT1 - 1>{}, std::true_type{});
      } else {
        if (last_tile) expand(std::integral_constant<int, NT1>{}, std::false_type{});
        else expand(std::integral_constant<int, NT1 - 1>{}, std::false_type{});
      }
    }
    MV_IR_STAMP();  // +1: expansion done
    if constexpr (EXPAND) __syncthreads();
    MV_IR_STAMP();  // +2

    // ---- phase 2: depthwise; item = (channel: the fastest index, segment of 8 outputs, output row)
    auto depthwise = [&](auto fma_mode) {
      constexpr bool FMA = decltype(fma_mode)::value;
      constexpr int ITEMS = 32 * NSEG * ORH;
      constexpr int NV = 8 * STRIDE;  // aligned input columns ix0 .. ix0 + NV - 1 of a segment
      const float* const zrow = wds + 288;
      for (int it = tid; it < ITEMS; it += NTHR) {
        const int c = it & 31, t2 = it >> 5;
        const int seg = t2 % NSEG, oyl = t2 / NSEG;
        const int oy = oy0 + oyl, ox0 = seg * 8, ix0 = ox0 * STRIDE;
        float wk[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) wk[i] = wds[c * 9 + i];
        const float na = t2a[c], nb = t2b[c];
        float rows[3][NV + 2];  // [0]: column ix0 - 1, [1 + i]: column ix0 + i, [NV + 1]: column ix0 + NV (stride 1 only)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = oy * STRIDE - 1 + ky;
          const float* rp = (iy >= 0 && iy < H) ? hid + c * HP + (iy - iy_lo) * W : zrow;
#pragma unroll
          for (int i = 0; i < NV / 4; ++i) {
            f32x4 q = zero4;
            if (W % NV == 0 || ix0 + 4 * i < W) q = *reinterpret_cast<const f32x4*>(rp + ix0 + 4 * i);
            rows[ky][4 * i + 1] = q.x, rows[ky][4 * i + 2] = q.y, rows[ky][4 * i + 3] = q.z, rows[ky][4 * i + 4] = q.w;
          }
          rows[ky][0] = ix0 > 0 ? rp[ix0 - 1] : 0.f;
          rows[ky][NV + 1] = (STRIDE == 1 && ix0 + NV < W) ? rp[ix0 + NV] : 0.f;
        }
        float* op = dws + (oyl * OW + ox0) * kOP + (c & 1) * 16 + (c >> 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (OW % 8 != 0 && ox0 + j >= OW) break;
          float a = fmaf(wk[0], rows[0][j * STRIDE], 0.f);
          a = fmaf(wk[1], rows[0][j * STRIDE + 1], a);
          a = fmaf(wk[2], rows[0][j * STRIDE + 2], a);
          a = fmaf(wk[3], rows[1][j * STRIDE], a);
          a = fmaf(wk[4], rows[1][j * STRIDE + 1], a);
          a = fmaf(wk[5], rows[1][j * STRIDE + 2], a);
          a = fmaf(wk[6], rows[2][j * STRIDE], a);
          a = fmaf(wk[7], rows[2][j * STRIDE + 1], a);
          a = fmaf(wk[8], rows[2][j * STRIDE + 2], a);
          op[j * kOP] = ir_relu6(ir_norm_t<FMA>(a, na, nb));
        }
      }
    };
    if (A.affine == 2) depthwise(std::true_type{}); else depthwise(std::false_type{});
    MV_IR_STAMP();  // +3: depthwise done
    __syncthreads();
    MV_IR_STAMP();  // +4

    // ---- phase 3: projection partial of the chunk (tiles past the region multiply rows nobody stores)
    if (cg == 0) {
      const float* ap = dws + (pg * 32 + l31) * kOP + hf * 16;
      const float* bp = w2s + l31 * kOP + hf * 16;
      int aoff[PTW];
#pragma unroll
      for (int i = 0; i < PTW; ++i) aoff[i] = min(i * PG, PTOUT - 1 - pg) * (32 * kOP);
      f32x4 aq[2][PTW], bq[2];
#pragma unroll
      for (int i = 0; i < PTW; ++i) aq[0][i] = *reinterpret_cast<const f32x4*>(ap + aoff[i]);
      bq[0] = *reinterpret_cast<const f32x4*>(bp);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (g + 1 < 4) {
#pragma unroll
          for (int i = 0; i < PTW; ++i) aq[(g + 1) & 1][i] = *reinterpret_cast<const f32x4*>(ap + aoff[i] + 4 * (g + 1));
          bq[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(bp + 4 * (g + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int i = 0; i < PTW; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[g & 1][i][e], bq[g & 1][e], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    MV_IR_STAMP();  // +5: projection of the chunk done
  }

  // ---- epilogue: through a wave-private [channel][pixel] buffer, 8 lanes per channel row of 32 pixels
  __syncthreads();
  float* const tb = hid + wave * (32 * kTB);
  const size_t plane = (size_t)OW * OW;
  const bool final_pass = A.slices == 1;
  float* const dst = final_pass ? A.y : A.part + (size_t)slice * A.n * A.cout * plane;
  const int r0 = lane >> 3, q4 = (lane & 7) * 4;
  if (cg == 0) {
#pragma unroll
    for (int i = 0; i < PTW; ++i) {
      const int pt = pg + PG * i;
      if (pt >= PTOUT) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(tb + l31 * kTB + 8 * g + 4 * hf) = (f32x4){acc[i][4 * g], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]};
      const int q = pt * 32 + q4;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const int co = r0 + 8 * jj;
        const f32x4 t4 = *reinterpret_cast<const f32x4*>(tb + co * kTB + q4);
        if (q >= OPI || co >= A.cout) continue;
        float v[4] = {t4.x, t4.y, t4.z, t4.w};
        if (final_pass) {
          const float na = A.a3[co], nb = A.b3[co];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ir_norm(v[r], na, nb, A.affine);
        }
        const size_t o = ((size_t)img * A.cout + co) * plane + (size_t)oy0 * OW + q;
        if (final_pass && A.res) {
          const f32x4 rv = *reinterpret_cast<const f32x4*>(A.res + o);
          v[0] = rv.x + v[0], v[1] = rv.y + v[1], v[2] = rv.z + v[2], v[3] = rv.w + v[3];
        }
        *reinterpret_cast<f32x4*>(dst + o) = (f32x4){v[0], v[1], v[2], v[3]};
      }
    }
  }
  MV_IR_STAMP();  // last: outputs stored
  }  // regions
}

// y = norm(part[0] + part[1] + ... (ascending)) [+ res]
struct IrReduceArgs {
  const float* part;
  const float *a3, *b3, *res;
  float* y;
  long long total, per_slice;  // n * cout * plane
  int plane, cout, slices, affine;
};

// The slices' partial sums are LOADED in batches of eight before they are added (in ascending order): written as one loop of
// load-then-add, every slice cost a memory round trip -- 20 us for the 30 slices of a 7 x 7 block at batch 1, more than the
// fused kernel in front of it (profiles/r03_ktrace_mobilenet_v2_b1.log).
__global__ __launch_bounds__(256) void k_invres_reduce(const IrReduceArgs A) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= A.total) return;
  constexpr int kBatch = 8;
  if ((A.plane & 3) == 0) {  // 4 consecutive outputs share their channel
    const int co = (int)((i4 / A.plane) % A.cout);
    f32x4 t = *reinterpret_cast<const f32x4*>(A.part + i4);
    for (int s0 = 1; s0 < A.slices; s0 += kBatch) {
      f32x4 p[kBatch];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) p[j] = *reinterpret_cast<const f32x4*>(A.part + (size_t)min(s0 + j, A.slices - 1) * A.per_slice + i4);
#pragma unroll
      for (int j = 0; j < kBatch; ++j)
        if (s0 + j < A.slices) t.x = t.x + p[j].x, t.y = t.y + p[j].y, t.z = t.z + p[j].z, t.w = t.w + p[j].w;
    }
    const float na = A.a3[co], nb = A.b3[co];
    float v[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = ir_norm(v[r], na, nb, A.affine);
    if (A.res) {
      const f32x4 rv = *reinterpret_cast<const f32x4*>(A.res + i4);
      v[0] = rv.x + v[0], v[1] = rv.y + v[1], v[2] = rv.z + v[2], v[3] = rv.w + v[3];
    }
    *reinterpret_cast<f32x4*>(A.y + i4) = (f32x4){v[0], v[1], v[2], v[3]};
  } else {  // 7 x 7 planes: 4 consecutive outputs may belong to two channels; dword-aligned 16-byte accesses where all four exist
    const bool whole = i4 + 3 < A.total;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    auto load4 = [&](const float* base) -> f32x4 {
      if (whole) {
        const f32x4u q = *reinterpret_cast<const f32x4u*>(base + i4);
        return (f32x4){q.x, q.y, q.z, q.w};
      }
      f32x4 q = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < 4 && i4 + r < A.total; ++r) q[r] = base[i4 + r];
      return q;
    };
    t = load4(A.part);
    for (int s0 = 1; s0 < A.slices; s0 += kBatch) {
      f32x4 p[kBatch];
#pragma unroll
      for (int j = 0; j < kBatch; ++j) p[j] = load4(A.part + (size_t)min(s0 + j, A.slices - 1) * A.per_slice);
#pragma unroll
      for (int j = 0; j < kBatch; ++j)
        if (s0 + j < A.slices) t.x = t.x + p[j].x, t.y = t.y + p[j].y, t.z = t.z + p[j].z, t.w = t.w + p[j].w;
    }
    for (int r = 0; r < 4 && i4 + r < A.total; ++r) {
      const long long i = i4 + r;
      const int co = (int)((i / A.plane) % A.cout);
      float v = ir_norm(t[r], A.a3[co], A.b3[co], A.affine);
      if (A.res) v = A.res[i] + v;
      A.y[i] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------- host side
struct IrGeom {
  bool ok;
  int imgs, strips, orows, rh_max, npin_max, npout_max, ptout;
  int slices, cps;
  int hp, off_hid, off_dws, off_w2, off_terms;
  size_t lds_bytes;
};

static int round4(int v) { return (v + 3) & ~3; }
static int pitch_b128(int v) {  // a multiple of 4 whose quarter is odd: 8 lanes' 16-byte LDS stores at this pitch hit 32 distinct banks
  v = round4(v);
  return (v / 4) % 2 ? v : v + 4;
}

// waves per workgroup of k_invres for a block shape (tuning build: MV_IR_WAVES forces 4 or 8)
static int ir_waves(int w, int stride, int cin, int cout) {
  if (const char* e = tune_env("MV_IR_WAVES")) return atoi(e) == 8 ? 8 : 4;
  (void)cin, (void)cout;
  // tools/sweep_ir_waves.py (profiles/r03_sweep_ir_waves.log): 8 waves are 4-9 % ahead on the 28-pixel blocks and the stride-1
  // 14-pixel ones (batch 1 and 64 alike), level on 14 -> 7, 10-20 % behind on the 7 x 7 blocks (4 tiles of 98 pixels for 8 waves)
  return (w == 28 || (w == 14 && stride == 1)) ? 8 : 4;
}

static IrGeom ir_geometry(int64_t n, int cin, int hidden, int cout, int h, int w, int stride) {
  IrGeom g = {};
  if (n <= 0 || h != w || (w != 7 && w != 14 && w != 28) || (stride != 1 && stride != 2) || (w == 7 && stride != 1)) return g;
  if (cin % 32 || hidden % 32 || cout % 32 || cin > 160 || cout > 320 || hidden < 32) return g;
  const int ow = (w - 1) / stride + 1, oh = ow;
  if (w == 28) {
    g.imgs = 1, g.orows = 7, g.strips = oh / 7;  // 28 -> 28: 4 strips of 7 rows; 28 -> 14: 2 strips of 7 rows
  } else if (w == 14) {
    g.imgs = 1, g.orows = oh, g.strips = 1;
  } else {
    g.imgs = 2, g.orows = 7, g.strips = 1;
  }
  g.rh_max = (g.strips == 1) ? h : (g.orows - 1) * stride + 3;  // strips in the middle of the image see one halo row above and below
  if (g.rh_max > h) g.rh_max = h;
  g.npin_max = g.imgs * g.rh_max * w;
  g.npout_max = g.imgs * g.orows * ow;
  g.ptout = (g.npout_max + 31) / 32;
  // slices: one workgroup per CU holds the LDS of a region, so a launch's time is (chunks per workgroup) x (time per chunk) as
  // long as it has at most 256 workgroups: as many slices as that allows (batch 64, 7 x 7: 32 regions -> 8 slices of 4 chunks)
  const int chunks = hidden / kHC;
  const long long regions = ((n + g.imgs - 1) / g.imgs) * g.strips;
  int want = (int)(256 / regions);
  if (const char* e = tune_env("MV_IR_SLICES")) want = atoi(e) > 0 ? atoi(e) : want;
  if (want < 1) want = 1;
  if (want > chunks) want = chunks;
  g.cps = (chunks + want - 1) / want;
  g.slices = (chunks + g.cps - 1) / g.cps;
  // LDS layout (floats): w1s [32][W1P] | hid [32][hp] | dws [32 * ptout][36] | w2s [cout][36] | terms
  const int w1p = (cin / 4) % 2 ? cin : cin + 4;
  g.hp = pitch_b128(32 * ((g.npin_max + 31) / 32));  // whole 32-pixel tiles: phase 1 stores a tile's rows without per-pixel bounds
  int off = round4(kHC * w1p);
  const int tbuf = ir_waves(w, stride, cin, cout) * 32 * 36;  // the epilogue's per-wave transpose buffers live in the dead hidden tile
  g.off_hid = off, off += kHC * g.hp > tbuf ? kHC * g.hp : tbuf;
  g.off_dws = off, off += 32 * g.ptout * 36;
  g.off_w2 = off, off += cout * 36;
  g.off_terms = off, off += 128 + 288 + 32;  // norm terms, depthwise taps, a row of zeros (the depthwise conv's padding rows)
  g.lds_bytes = (size_t)off * sizeof(float);
  g.ok = g.lds_bytes <= 160 * 1024;
  return g;
}

// wide maps (k_invres_wide): output rows per region by (map side, stride)
// (output rows per region, waves per workgroup, waves per SIMD the registers are bounded for): the variants built per block shape;
// index 0 is the product's choice, the others are reachable through MV_IRW_VARIANT in the tuning build (tools/perf_invres.py)
struct IrwVariant { int orows, waves, occ; };
static const IrwVariant kIrw112[] = {{2, 8, 2}, {1, 4, 2}, {1, 8, 2}, {2, 4, 1}, {2, 16, 4}};
static const IrwVariant kIrw56s1[] = {{7, 8, 2}, {2, 4, 2}, {2, 8, 2}, {4, 8, 2}, {4, 4, 1}, {7, 16, 4}};
static const IrwVariant kIrw56s2[] = {{4, 8, 2}, {2, 4, 2}, {2, 8, 2}, {2, 4, 1}, {1, 4, 2}, {7, 8, 2}, {7, 16, 4}};
static int irw_variant_index(int count) {
  const char* e = tune_env("MV_IRW_VARIANT");
  const int v = (e && *e) ? atoi(e) : 0;
  return v >= 0 && v < count ? v : 0;
}
static const IrwVariant kIrw112s1[] = {{4, 8, 2}, {2, 8, 2}, {4, 16, 4}, {2, 4, 2}};
static IrwVariant irw_variant(int w, int stride) {
  if (w == 112 && stride == 1) return kIrw112s1[irw_variant_index(4)];
  if (w == 112) return kIrw112[irw_variant_index(5)];
  return stride == 1 ? kIrw56s1[irw_variant_index(6)] : kIrw56s2[irw_variant_index(7)];
}
static int irw_orows(int w, int stride) { return irw_variant(w, stride).orows; }

static bool irw_shape(int cin, int hidden, int cout, int h, int w, int stride) {
  if (h != w || hidden < 32 || hidden % 8 || cout < 1 || cout > 32) return false;
  if (w == 112 && stride == 1) return cin == 32 && hidden == 32;  // MobileNetV2's first block: expand_ratio 1, no expansion conv
  if (w == 112) return stride == 2 && cin == 16;
  if (w == 56) return cin == 24;
  return false;
}

static IrGeom irw_geometry(int64_t n, int cin, int hidden, int cout, int h, int w, int stride) {
  IrGeom g = {};
  if (n <= 0 || (stride != 1 && stride != 2) || !irw_shape(cin, hidden, cout, h, w, stride)) return g;
  const int ow = w / stride;
  g.imgs = 1, g.orows = irw_orows(w, stride), g.strips = ow / g.orows;
  g.rh_max = (g.orows - 1) * stride + 3;
  g.npin_max = g.rh_max * w;
  g.npout_max = g.orows * ow;
  g.ptout = (g.npout_max + 31) / 32;
  const int w1p = (cin / 4) % 2 ? cin : cin + 4;
  g.hp = pitch_b128(32 * ((g.npin_max + 31) / 32));  // (hidden == cin: the flat copy of the input rows needs rh_max * w <= hp, the same bound)
  const IrwVariant var = irw_variant(w, stride);
  const int tbuf = var.waves * 32 * 36;  // the epilogue's per-wave transpose buffers live in the dead hidden tile
  int off = hidden == cin ? 0 : round4(kHC * w1p);  // no expansion: no expand weights
  g.off_hid = off, off += kHC * g.hp > tbuf ? kHC * g.hp : tbuf;
  g.off_dws = off, off += 32 * g.ptout * 36;
  g.off_w2 = off, off += 32 * 36;
  g.off_terms = off, off += 128 + 288 + round4(w + 4);
  g.lds_bytes = (size_t)off * sizeof(float);
  // slices: only when the launch would leave CUs without a workgroup (a CU holds as many as its LDS and 16 wave slots allow)
  const int chunks = (hidden + kHC - 1) / kHC;
  const long long regions = n * g.strips;
  int per_cu = (int)(160 * 1024 / g.lds_bytes);
  if (per_cu > 16 / var.waves) per_cu = 16 / var.waves;
  if (per_cu < 1) per_cu = 1;
  int want = (int)(256 * per_cu / regions);
  if (const char* e = tune_env("MV_IR_SLICES")) want = atoi(e) > 0 ? atoi(e) : want;
  if (want < 1) want = 1;
  if (want > chunks) want = chunks;
  g.cps = (chunks + want - 1) / want;
  g.slices = (chunks + g.cps - 1) / g.cps;
  g.ok = g.lds_bytes <= 160 * 1024;
  return g;
}

template <int W, int STRIDE, int ORH, int CIN, int NW, int OCC, bool EXPAND = true>
static int irw_launch(const IrArgs& a, const IrGeom& g, unsigned regions, hipStream_t s) {
  auto kern = k_invres_wide<W, STRIDE, ORH, CIN, NW, OCC, EXPAND>;
  static int lds_limit = 0;  // per instantiation; raised once (a benign race: two first callers set the same value)
  if (lds_limit < (int)g.lds_bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
    lds_limit = (int)g.lds_bytes;
  }
  // persistent workgroups: as many as the chip holds at once (per_cu per CU), each walking an equal share of the regions
  int per_cu = (int)(160 * 1024 / g.lds_bytes);
  if (per_cu > 16 / NW) per_cu = 16 / NW;
  if (per_cu < 1) per_cu = 1;
  const unsigned slots = (unsigned)(256 * per_cu / g.slices > 0 ? 256 * per_cu / g.slices : 1);
  const unsigned rounds = (regions + slots - 1) / slots;
  const unsigned grid = (regions + rounds - 1) / rounds;
  hipLaunchKernelGGL(kern, dim3(grid, (unsigned)g.slices), dim3(64 * NW), g.lds_bytes, s, a);
  return check_launchf("k_invres_wide<%d,s%d,rows%d,cin%d,waves%d,slices%d>%s", W, STRIDE, ORH, CIN, NW, g.slices, EXPAND ? "" : " (no expansion)");
}

template <int W, int STRIDE, int PTOUT, int COT, int CINQ, int NW>
static int ir_launch_nw(const IrArgs& a, const IrGeom& g, unsigned regions, hipStream_t s) {
  auto kern = k_invres<W, STRIDE, PTOUT, COT, CINQ, NW>;
  static int lds_limit = 0;  // per instantiation; raised once (a benign race: two first callers set the same value)
  if (lds_limit < (int)g.lds_bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
    lds_limit = (int)g.lds_bytes;
  }
  hipLaunchKernelGGL(kern, dim3(regions, (unsigned)g.slices), dim3(64 * NW), g.lds_bytes, s, a);
  return check_launchf("k_invres<%d,s%d,cin%d,cout%d,slices%d>%s", W, STRIDE, 32 * CINQ, 32 * COT, g.slices, NW == 8 ? " (8 waves)" : "");
}
template <int W, int STRIDE, int PTOUT, int COT, int CINQ>
static int ir_launch(const IrArgs& a, const IrGeom& g, unsigned regions, hipStream_t s) {
  if (ir_waves(W, STRIDE, 32 * CINQ, 32 * COT) == 8) return ir_launch_nw<W, STRIDE, PTOUT, COT, CINQ, 8>(a, g, regions, s);
  return ir_launch_nw<W, STRIDE, PTOUT, COT, CINQ, 4>(a, g, regions, s);
}

// the instantiations: MobileNetV2's blocks on 28 / 14 / 7-pixel maps (width multiplier 1.0, 224 x 224 input)
static bool ir_instantiated(int w, int stride, int cin, int cout) {
  if (w == 28 && stride == 1) return cin == 32 && cout == 32;
  if (w == 28 && stride == 2) return cin == 32 && cout == 64;
  if (w == 14 && stride == 1) return (cin == 64 && (cout == 64 || cout == 96)) || (cin == 96 && cout == 96);
  if (w == 14 && stride == 2) return cin == 96 && cout == 160;
  if (w == 7 && stride == 1) return cin == 160 && (cout == 160 || cout == 320);
  return false;
}

// the geometry of whichever kernel covers the shape (ok == false: none)
static IrGeom ir_any_geometry(int64_t n, int cin, int hidden, int cout, int h, int w, int stride, bool* wide) {
  IrGeom g = irw_geometry(n, cin, hidden, cout, h, w, stride);
  *wide = g.ok;
  if (g.ok) return g;
  g = ir_geometry(n, cin, hidden, cout, h, w, stride);
  if (g.ok && !ir_instantiated(w, stride, cin, cout)) g.ok = false;
  return g;
}

int invres_plan(int64_t n, int cin, int hidden, int cout, int h, int w, int stride, int* slices, int* slice_len) {
  bool wide = false;
  const IrGeom g = ir_any_geometry(n, cin, hidden, cout, h, w, stride, &wide);
  if (!g.ok) return 0;
  *slices = g.slices;
  *slice_len = g.slices > 1 ? g.cps * kHC : hidden;
  return 1;
}

int64_t invres_workspace_bytes(int64_t n, int cin, int hidden, int cout, int h, int w, int stride) {
  bool wide = false;
  const IrGeom g = ir_any_geometry(n, cin, hidden, cout, h, w, stride, &wide);
  if (!g.ok || g.slices == 1) return 0;
  const int ow = (w - 1) / stride + 1;
  return (int64_t)g.slices * n * cout * ow * ow * (int64_t)sizeof(float);
}

int launch_invres(const float* x, const float* w1, const float* a1, const float* b1, const float* wd, const float* a2, const float* b2,
                  const float* w2, const float* a3, const float* b3, int residual, float* y, int64_t n, int cin, int hidden, int cout,
                  int h, int w, int stride, int affine, void* workspace, int64_t workspace_bytes, hipStream_t s) {
  bool wide = false;
  const IrGeom g = ir_any_geometry(n, cin, hidden, cout, h, w, stride, &wide);
  if (!g.ok)
    return set_error(MV_ERR_UNSUPPORTED, "inverted_residual: no fused kernel for %d -> %d -> %d on a %d x %d map, stride %d "
                     "(mv_inverted_residual_k_slices() returns 0: run the block as three mv_conv_norm_act_f32 calls)", cin, hidden, cout, h, w, stride);
  if (residual && (stride != 1 || cin != cout)) return set_error(MV_ERR_INVALID_ARGUMENT, "inverted_residual: `+ x` needs stride 1 and cin == cout");
  const int ow = (w - 1) / stride + 1;
  const int64_t need = g.slices > 1 ? (int64_t)g.slices * n * cout * ow * ow * (int64_t)sizeof(float) : 0;
  if (need && (!workspace || workspace_bytes < need))
    return set_error(MV_ERR_INVALID_ARGUMENT, "inverted_residual: this shape sums the projection in %d slices and needs a workspace of %lld bytes "
                     "(mv_inverted_residual_workspace_bytes)", g.slices, (long long)need);
  if (hidden != cin && (!w1 || !a1 || !b1)) return set_error(MV_ERR_INVALID_ARGUMENT, "inverted_residual: the expansion's weights / norm may be null only for a block without expansion (hidden == cin)");
  if ((uintptr_t)x % 16 || (uintptr_t)y % 16 || (uintptr_t)w1 % 16 || (uintptr_t)w2 % 16 || (need && (uintptr_t)workspace % 16))
    return set_error(MV_ERR_INVALID_ARGUMENT, "inverted_residual: x, y, the 1x1 weights and the workspace must be 16-byte aligned");
  const long long regions = ((n + g.imgs - 1) / g.imgs) * g.strips;
  if (regions > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "inverted_residual: batch too large for one launch");
  IrArgs a = {};
  a.x = x, a.w1 = w1, a.a1 = a1, a.b1 = b1, a.wd = wd, a.a2 = a2, a.b2 = b2, a.w2 = w2, a.a3 = a3, a.b3 = b3;
  a.res = residual ? x : nullptr, a.y = y, a.part = static_cast<float*>(workspace);
  a.n = (int)n, a.hidden = hidden, a.cout = cout, a.affine = affine;
  a.imgs = g.imgs, a.strips = g.strips, a.slices = g.slices, a.cps = g.cps;
  a.hp = g.hp, a.off_hid = g.off_hid, a.off_dws = g.off_dws, a.off_w2 = g.off_w2, a.off_terms = g.off_terms;
  int rc = MV_ERR_UNSUPPORTED;
  const unsigned r = (unsigned)regions;
  if (wide) {
    const IrwVariant v = irw_variant(w, stride);
#define MV_IRW_CASE(W_, S_, CIN_, R_, NW_, OCC_) \
    if (w == W_ && stride == S_ && v.orows == R_ && v.waves == NW_ && v.occ == OCC_) rc = irw_launch<W_, S_, R_, CIN_, NW_, OCC_>(a, g, r, s)
    MV_IRW_CASE(112, 2, 16, 2, 4, 1); MV_IRW_CASE(112, 2, 16, 1, 4, 2); MV_IRW_CASE(112, 2, 16, 1, 8, 2); MV_IRW_CASE(112, 2, 16, 2, 8, 2);
    if (w == 112 && stride == 1) {
      if (v.orows == 4 && v.waves == 8) rc = irw_launch<112, 1, 4, 32, 8, 2, false>(a, g, r, s);
      else if (v.orows == 2 && v.waves == 8) rc = irw_launch<112, 1, 2, 32, 8, 2, false>(a, g, r, s);
      else if (v.orows == 4) rc = irw_launch<112, 1, 4, 32, 16, 4, false>(a, g, r, s);
      else rc = irw_launch<112, 1, 2, 32, 4, 2, false>(a, g, r, s);
    }
    MV_IRW_CASE(112, 2, 16, 2, 16, 4); MV_IRW_CASE(56, 1, 24, 7, 16, 4); MV_IRW_CASE(56, 2, 24, 7, 8, 2); MV_IRW_CASE(56, 2, 24, 7, 16, 4);
    MV_IRW_CASE(56, 1, 24, 4, 4, 1); MV_IRW_CASE(56, 1, 24, 2, 4, 2); MV_IRW_CASE(56, 1, 24, 2, 8, 2); MV_IRW_CASE(56, 1, 24, 4, 8, 2); MV_IRW_CASE(56, 1, 24, 7, 8, 2);
    MV_IRW_CASE(56, 2, 24, 2, 4, 1); MV_IRW_CASE(56, 2, 24, 2, 4, 2); MV_IRW_CASE(56, 2, 24, 2, 8, 2); MV_IRW_CASE(56, 2, 24, 4, 8, 2); MV_IRW_CASE(56, 2, 24, 1, 4, 2);
#undef MV_IRW_CASE
  }
  else if (w == 28 && stride == 1) rc = ir_launch<28, 1, 7, 1, 1>(a, g, r, s);
  else if (w == 28) rc = ir_launch<28, 2, 4, 2, 1>(a, g, r, s);
  else if (w == 14 && stride == 1 && cin == 64 && cout == 64) rc = ir_launch<14, 1, 7, 2, 2>(a, g, r, s);
  else if (w == 14 && stride == 1 && cin == 64) rc = ir_launch<14, 1, 7, 3, 2>(a, g, r, s);
  else if (w == 14 && stride == 1) rc = ir_launch<14, 1, 7, 3, 3>(a, g, r, s);
  else if (w == 14) rc = ir_launch<14, 2, 2, 5, 3>(a, g, r, s);
  else if (cout == 160) rc = ir_launch<7, 1, 4, 5, 5>(a, g, r, s);
  else rc = ir_launch<7, 1, 4, 10, 5>(a, g, r, s);
  if (rc != MV_OK || g.slices == 1) return rc;
  IrReduceArgs ra = {};
  ra.part = a.part, ra.a3 = a3, ra.b3 = b3, ra.res = a.res, ra.y = y;
  ra.plane = ow * ow, ra.cout = cout, ra.slices = g.slices, ra.affine = affine;
  ra.per_slice = (long long)n * cout * ra.plane, ra.total = ra.per_slice;
  const long long blocks = (ra.total / 4 + 255 + 1) / 256;
  hipLaunchKernelGGL(k_invres_reduce, dim3((unsigned)blocks), dim3(256), 0, s, ra);
  return check_launchf("k_invres%s<%d,s%d,cin%d,cout%d,slices%d>+k_invres_reduce", wide ? "_wide" : "", w, stride, cin, cout, g.slices);
}

}  // namespace mv
