// conv3x3_mfma.hip -- nn.Conv2d(cin, cout, 3, padding=1) [+bias] [+ReLU] as an implicit GEMM on the
// gfx950 fp32 matrix core.
// WHICH SHAPES STILL REACH THIS KERNEL (round 2): the first layer runs on k_conv3x3_c3 (conv3x3_c3.hip), every other
// 3x3 layer on k_conv3x3_gen (conv3x3_gen.hip), whose LDS tile holds whole rows up to 510 pixels.  k_conv3x3 here is what
// serves (a) feature maps WIDER than 510 pixels with cin != 3 (or cin == 3 with W % 4 != 0 / an unaligned output), and
// (b) images whose output plane set exceeds 4 GB (32-bit per-image byte offsets in the other two) -- mv_conv3x3_bias_relu_f32
// in abi.hip tries c3, then gen, then this.  tests/test_gpu_cnn.py::test_wide_maps_take_the_first_generation_kernel.  Replaces `conv2d = nn.Conv2d(in_channels, v, kernel_size=3, padding=1)`
// + `nn.ReLU(inplace=True)` of make_layers (models/vgg.py:81-85) and Conv2dNormActivation with
// norm_layer=None (ops/misc.py:97-119).  BASELINE cfg4: 256 x (3,224,224) -> (64,224,224).
//
// This path genuinely is a dense im2col x weight contraction (M = cout, N = pixels, K = cin*9), the one
// place in the hot path where MFMA belongs:
//   * v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate, bit-for-bit a k-ordered fmaf chain -- so the
//     result equals oracle/oracle.c's (ci,dy,dx)-ordered chain exactly; no reduced precision.
//   * im2col is never materialised (the reference's deform_conv2d writes a `columns` buffer,
//     csrc/ops/cpu/deform_conv2d_kernel.cpp:118-193): the B fragment of k-step s is ONE ds_read_b32
//     per lane straight out of the zero-padded input tile in LDS (lanes 0-31 take k = 2s, lanes 32-63
//     take k = 2s+1; 32 consecutive pixels per half: conflict-free).
//   * the A fragments (weights) are laid out in LDS in fragment order once per workgroup; for cin = 3
//     (K = 27 -> 14 k-steps, 28 VGPRs for both 32-channel tiles) they then live in registers.
//   * orientation: channels on the accumulator rows, pixels on the lanes, so every accumulator
//     register stores two 128-byte runs of 32 consecutive pixels of one channel plane (NCHW,
//     W fastest), with bias + ReLU fused into the store.  The op is HBM-WRITE-bound (AI 12.9 FLOP/B):
//     44.4 GFLOP of MFMA take 282 us at peak, 3.29 GB of output take >= 430 us.
#include "mv_common.h"
#include "mv_act.h"

namespace mv {

#ifndef MV_CONV_NT
#define MV_CONV_NT 1
#endif
#ifndef MV_CONV_ABLATE_STORE
#define MV_CONV_ABLATE_STORE 0  // profiling builds only: results are wrong
#endif
#ifndef MV_CONV_ABLATE_MFMA
#define MV_CONV_ABLATE_MFMA 0
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
  const float* x;
  const float* w;
  const float* b;
  float* y;
  int cin, cout, h, wdt;
  int K;        // cin * 9
  int ksteps;   // ceil((K + 1) / 2): k = K is the bias slot (A = bias[co], B = 1)
  int mtiles;   // ceil(cout / 32)
  int th;       // output rows per workgroup
  int wc;       // output columns per workgroup (multiple of 32, <= 256)
  int ntx;      // wc / 32
  int pitch;    // LDS row pitch of the input tile (wc + 2, padded)
  int tiles_x, tiles_y;
  int relu;
  int vec_rows;  // every input row starts 16-byte aligned and w % 4 == 0
  unsigned nblocks;
};

template <bool RELU>
__device__ inline float act(float v) {
  if (RELU) v = relu_f32(v);  // NaN passes through, like torch.relu
  return v;
}

// KS > 0: static k-step count, weights register-resident (requires mtiles <= MT).
// KS == 0: runtime k-steps, weights re-read from LDS per step, any number of 32-channel tiles.
// FULLM: cout is a multiple of 32 (no per-channel store predicate).
// FULLW: w is a multiple of 32 (no per-lane store predicate: the pipelined loop is one basic block).
template <int KS, int MT, bool FULLM, bool RELU, bool FULLW>
__global__ __launch_bounds__(256, 2) void k_conv3x3(const ConvArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: tile index math runs on the SALU
  const int l31 = lane & 31, hf = lane >> 5;
  const int cin = A.cin, cout = A.cout, h = A.h, w = A.wdt;
  const int th = A.th, wc = A.wc, pitch = A.pitch, K = A.K;
  const int ksteps = KS ? KS : A.ksteps;
  const int tile_rows = th + 2;
  float* xin = lds;                                   // [cin][th+2][pitch]
  float* wfr = lds + cin * tile_rows * pitch;         // [mtiles][ksteps][64]   A fragments
  int* koff = reinterpret_cast<int*>(wfr + A.mtiles * ksteps * 64);  // [2*ksteps] tile offset of tap k

  const unsigned wid = xcd_remap(blockIdx.x, A.nblocks);
  const int tx = wid % A.tiles_x;
  const unsigned t2 = wid / A.tiles_x;
  const int ty = t2 % A.tiles_y;
  const long long img = t2 / A.tiles_y;
  const int xb = tx * wc, yb = ty * th;
  const float* xp = A.x + (size_t)img * cin * h * w;

  // ---- A fragments in fragment order: wfr[(m*ksteps + s)*64 + l] = W[32m + (l&31)][2s + (l>>5)]
  for (int q = wave; q < A.mtiles * ksteps; q += 4) {  // q is wave-uniform: the divisions are scalar
    const int s = q % ksteps, m = q / ksteps;
    const int co = 32 * m + l31, k = 2 * s + hf;
    float v = 0.f;
    if (co < cout && k < K) v = A.w[(size_t)co * K + k];
    if (co < cout && k == K && A.b != nullptr) v = A.b[co];  // bias rides the chain: fma(bias, 1, acc) == acc + bias
    wfr[q * 64 + lane] = v;
  }
  // ---- tap k = (ci, dy, dx) -> offset inside the input tile
  for (int k = tid; k < 2 * ksteps; k += 256) {
    const int kk = k < K ? k : 0;
    const int ci = kk / 9, r = kk - 9 * ci, dy = r / 3, dx = r - 3 * dy;
    koff[k] = (ci * tile_rows + dy) * pitch + dx;
  }
  // ---- zero-padded input tile: rows yb-1 .. yb+th, columns xb-1 .. xb+wc.  Loads are issued in batches of
  //      kStage independent requests per lane before any of them is consumed (the naive load -> wait -> ds_write
  //      loop serialised one HBM round trip per element and cost a third of the kernel).
  {
    constexpr int kStage = 8;
    const int cols = wc + 2;
    const int rows_total = cin * tile_rows;
    const bool vec = A.vec_rows;  // rows 16-byte aligned: stage [xb-4, xb+wc+4) with one 16-byte load per lane
    if (vec) {
      const int nq = (wc + 8) >> 2;  // float4 slots per row (<= 66 -> two passes over the lanes at most)
      const int per_row = (nq + kWave - 1) / kWave;
      const int nitems = rows_total * per_row;
      for (int base = wave; base < nitems; base += 4 * kStage) {
        f32x4 v[kStage];
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
          const int it = base + 4 * u;
          v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (it < nitems) {
            const int r = it / per_row, q = (it - r * per_row) * kWave + lane;
            const int ci = r / tile_rows, rr = r - ci * tile_rows;
            const int gy = yb - 1 + rr, gx0 = xb - 4 + 4 * q;
            if (q < nq && gy >= 0 && gy < h && gx0 >= 0 && gx0 + 3 < w)
              v[u] = *reinterpret_cast<const f32x4*>(xp + ((size_t)ci * h + gy) * w + gx0);
          }
        }
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
          const int it = base + 4 * u;
          if (it < nitems) {
            const int r = it / per_row, q = (it - r * per_row) * kWave + lane;
            if (q < nq) {
              float* dst = xin + r * pitch + (4 * q - 3);  // tile column of gx0 = gx0 - (xb - 1)
              const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int c = 4 * q - 3 + j;
                if (c >= 0 && c < cols) dst[j] = e[j];
              }
            }
          }
        }
      }
    } else {
      const int per_row = (cols + kWave - 1) / kWave;
      const int nitems = rows_total * per_row;
      for (int base = wave; base < nitems; base += 4 * kStage) {
        float v[kStage];
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
          const int it = base + 4 * u;
          v[u] = 0.f;
          if (it < nitems) {
            const int r = it / per_row, c = (it - r * per_row) * kWave + lane;
            const int ci = r / tile_rows, rr = r - ci * tile_rows;
            const int gy = yb - 1 + rr, gx = xb - 1 + c;
            if (c < cols && gy >= 0 && gy < h && gx >= 0 && gx < w) v[u] = xp[((size_t)ci * h + gy) * w + gx];
          }
        }
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
          const int it = base + 4 * u;
          if (it < nitems) {
            const int r = it / per_row, c = (it - r * per_row) * kWave + lane;
            if (c < cols) xin[r * pitch + c] = v[u];
          }
        }
      }
    }
  }
  __syncthreads();
#if defined(MV_CONV_PROLOGUE_ONLY)
  if (xin[tid] == 12345.678f) A.y[tid] = 1.f;
  return;
#endif

  const int ntiles = th * A.ntx;
  const size_t plane = (size_t)h * w;
  // lane-constant part of the output address: channel offset 4*hf inside every 8-channel group
  float* const ybase = A.y + (size_t)img * cout * plane + (size_t)(4 * hf) * plane;

  auto epilogue = [&](const f32x16 (&acc)[MT], int m0, int oy, int px) {
    if (px < w) {
      float* yo = ybase + (size_t)oy * w + px;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (m0 + m >= A.mtiles) break;  // wave-uniform: this 32-channel tile does not exist
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int cu = 32 * (m0 + m) + (i & 3) + 8 * (i >> 2);  // + 4*hf (in ybase): wave-uniform part
          const float v = act<RELU>(acc[m][i]);
#if MV_CONV_ABLATE_STORE
          if (v == 12345.678f) yo[(size_t)cu * plane] = v;
#elif MV_CONV_NT
          if (FULLM || cu + 4 * hf < cout) __builtin_nontemporal_store(v, yo + (size_t)cu * plane);
#else
          if (FULLM || cu + 4 * hf < cout) yo[(size_t)cu * plane] = v;
#endif
        }
      }
    }
  };
  if constexpr (KS > 0) {
    // Software-pipelined over this wave's tiles (two accumulator sets):
    //   while the MFMA chain of tile j runs into one set, the SAME wave drains the other set (tile j-1):
    //   v_accvgpr_read + bias + ReLU + store, 2-3 values per k-step, and fetches the B fragments of tile j+1.
    // One wave therefore keeps its SIMD's matrix pipe busy by itself; without this the waves of a CU fall into
    // lockstep (all in their MFMA chains, then all in their epilogues) and chain / epilogue / stores add up
    // instead of overlapping (measured: 0.75 ms = 0.23 + 0.29 + store tail; tools/tune_dw3x3.py --op conv).
    static_assert(MT == 2, "pipelined path is written for two 32-channel tiles");
    float afr[MT][KS];
    int off[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      off[s] = koff[2 * s + hf];
#pragma unroll
      for (int m = 0; m < MT; ++m) afr[m][s] = (m < A.mtiles) ? wfr[(m * KS + s) * 64 + lane] : 0.f;
    }
    // last k-step: k = 2(KS-1)+hf is a real tap (k < K), the bias slot (k == K: B = 1) or padding (B = 0)
    const bool last_valid = (2 * (KS - 1) + hf) < K;
    const float last_fill = (2 * (KS - 1) + hf) == K ? 1.f : 0.f;
    // this wave's tiles are t = wave + 4j; the valid ones (output row inside the image) are a prefix
    const int rows_here = min(th, h - yb);
    const int nvalid = rows_here * A.ntx;
    const int nw = (nvalid > wave) ? (nvalid - wave + 3) / 4 : 0;
    auto tile_bp = [&](int j) -> const float* {
      const int t = wave + 4 * j, ly = t / A.ntx, nx = t - ly * A.ntx;
      return xin + ly * pitch + nx * 32 + l31;
    };
    // output address = uniform base (SGPR pair) + 32-bit per-lane byte offset -> global_store ... saddr form,
    // the per-channel stride is added on the scalar unit
    char* const simg = reinterpret_cast<char*>(A.y + (size_t)img * cout * plane);
    float sink = 0.f;  // only used by the MV_CONV_ABLATE_STORE == 3 profiling build
    (void)sink;
    auto tile_voff = [&](int j, bool& ok) -> unsigned {
      const int t = wave + 4 * j, ly = t / A.ntx, nx = t - ly * A.ntx;
      const int px = xb + nx * 32 + l31;
      ok = FULLW || px < w;
      return (unsigned)(((size_t)(4 * hf) * plane + (size_t)(yb + ly) * w + px) * sizeof(float));
    };
    auto drain = [&](const f32x16 (&acc)[MT], unsigned voff, bool ok, int v0, int v1) {  // values [v0, v1) of 32
#pragma unroll
      for (int v = 0; v < 32; ++v) {
        if (v >= v0 && v < v1) {
          const int m = v >> 4, i = v & 15;
          const int cu = 32 * m + (i & 3) + 8 * (i >> 2);
          const float r = act<RELU>(acc[m][i]);
          float* dst = reinterpret_cast<float*>(simg + (size_t)cu * plane * sizeof(float) + voff);
#if MV_CONV_ABLATE_STORE == 3
          // no store instructions at all: fold the value into a per-lane sink (kept alive, stored once at the end)
          sink += r;
          (void)dst;
#elif MV_CONV_ABLATE_STORE == 2
          // same instruction stream, but every store lands in one L2-resident 4 MiB window
          dst = reinterpret_cast<float*>(reinterpret_cast<char*>(A.y) + (((size_t)cu * plane * 4 + voff) & 0x3FFFFCu));
          *dst = r;
#elif MV_CONV_ABLATE_STORE
          if (r == 12345.678f) *dst = r;
#elif MV_CONV_NT
          if ((FULLW || ok) && (FULLM || cu + 4 * hf < cout)) __builtin_nontemporal_store(r, dst);
#else
          if ((FULLW || ok) && (FULLM || cu + 4 * hf < cout)) *dst = r;
#endif
        }
      }
    };
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    if (nw > 0) {
      f32x16 accA[MT], accB[MT];
      float bvA[KS], bvB[KS];
      {  // B fragments of tile 0
        const float* bp = tile_bp(0);
#pragma unroll
        for (int s = 0; s < KS; ++s) bvA[s] = bp[off[s]];
        bvA[KS - 1] = last_valid ? bvA[KS - 1] : last_fill;
      }
      // One pipeline stage: chain of tile j into `cur` from `bcur`; drain `prev` (tile j-1) if there is one;
      // fetch tile j+1's fragments into `bnext` (clamped to the last tile: a harmless re-read at the end).
#define MV_STAGE(cur, prev, bcur, bnext, j, have_prev)                                                   \
  {                                                                                                      \
    const float* bpn = tile_bp(min((j) + 1, nw - 1));                                                     \
    bool okp = false;                                                                                    \
    const unsigned vo = (have_prev) ? tile_voff((j)-1, okp) : 0u;                                         \
    _Pragma("unroll") for (int s = 0; s < KS; ++s) {                                                     \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                   \
        cur[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[m][s], bcur[s], s == 0 ? zero : cur[m], 0, 0, 0); \
        if (m == 0) bnext[s] = bpn[off[s]];                                                              \
        if (have_prev) drain(prev, vo, okp, (32 * (2 * s + m)) / (2 * KS), (32 * (2 * s + m + 1)) / (2 * KS)); \
      }                                                                                                  \
    }                                                                                                    \
    bnext[KS - 1] = last_valid ? bnext[KS - 1] : last_fill;                                                   \
  }
      MV_STAGE(accA, accB, bvA, bvB, 0, false)
      int j = 1;
      for (; j + 1 < nw; j += 2) {
        MV_STAGE(accB, accA, bvB, bvA, j, true)
        MV_STAGE(accA, accB, bvA, bvB, j + 1, true)
      }
      bool okl = false;
      if (j < nw) {  // one more (odd-indexed) tile, then drain it
        MV_STAGE(accB, accA, bvB, bvA, j, true)
        const unsigned vl = tile_voff(j, okl);
        drain(accB, vl, okl, 0, 32);
      } else {
        const unsigned vl = tile_voff(j - 1, okl);
        drain(accA, vl, okl, 0, 32);
      }
#undef MV_STAGE
#if MV_CONV_ABLATE_STORE == 3
      if (sink == 12345.678f) A.y[tid] = sink;
#endif
    }
  } else {
    for (int m0 = 0; m0 < A.mtiles; m0 += MT) {
      for (int t = wave; t < ntiles; t += 4) {
        const int ly = t / A.ntx, nx = t - ly * A.ntx;
        const int oy = yb + ly;
        if (oy >= h) break;
        const float* bp = xin + ly * pitch + nx * 32 + l31;
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
        for (int s = 0; s < ksteps; ++s) {
          const int k = 2 * s + hf;
          float bv = bp[koff[k]];
          bv = (k < K) ? bv : (k == K ? 1.f : 0.f);
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float av = (m0 + m < A.mtiles) ? wfr[((m0 + m) * ksteps + s) * 64 + lane] : 0.f;
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
          }
        }
        epilogue(acc, m0, oy, xb + nx * 32 + l31);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <int KS, int MT, bool FULLM, bool RELU, bool FULLW>
static int launch_kw(const ConvArgs& a, size_t lds_bytes, hipStream_t s) {
  auto k = k_conv3x3<KS, MT, FULLM, RELU, FULLW>;
  if (lds_bytes > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
  hipLaunchKernelGGL(k, dim3(a.nblocks), dim3(256), lds_bytes, s, a);
  return check_launch("k_conv3x3");
}

template <int KS, int MT, bool FULLM>
static int launch_k(const ConvArgs& a, size_t lds_bytes, hipStream_t s) {
  const bool fullw = KS > 0 && (a.wdt % 32 == 0);  // only the pipelined path distinguishes it
  if (fullw)
    return a.relu ? launch_kw<KS, MT, FULLM, true, true>(a, lds_bytes, s) : launch_kw<KS, MT, FULLM, false, true>(a, lds_bytes, s);
  return a.relu ? launch_kw<KS, MT, FULLM, true, false>(a, lds_bytes, s) : launch_kw<KS, MT, FULLM, false, false>(a, lds_bytes, s);
}

int launch_conv3x3(const float* x, const float* w, const float* b, float* y, int64_t n, int cin, int h, int wdt,
                   int cout, int relu, hipStream_t s) {
  ConvArgs a = {};
  a.x = x, a.w = w, a.b = b, a.y = y;
  a.cin = cin, a.cout = cout, a.h = h, a.wdt = wdt, a.relu = relu;
  a.K = cin * 9;
  a.vec_rows = (wdt % 4 == 0) && ((uintptr_t)x % 16 == 0);
  a.ksteps = (a.K + 2) / 2;
  a.mtiles = (cout + 31) / 32;
  a.wc = ((wdt + 31) / 32) * 32;
  if (a.wc > 256) a.wc = 256;
  a.ntx = a.wc / 32;
  a.pitch = a.wc + 2 + 1;  // +1: odd pitch keeps the two lane halves (rows dy, dy') off the same banks
  // rows per workgroup: as many as fit a 64 KiB budget (>= 2 workgroups per CU), at most 8
  const size_t fixed = ((size_t)a.mtiles * a.ksteps * 64 + 2 * a.ksteps) * sizeof(float);
  int th = 8;
  auto bytes = [&](int rows) { return (size_t)cin * (rows + 2) * a.pitch * sizeof(float) + fixed; };
  while (th > 1 && bytes(th) > 64 * 1024) th >>= 1;
  if (bytes(th) > 160 * 1024)
    return set_error(MV_ERR_UNSUPPORTED, "conv3x3: cin=%d cout=%d needs %zu B of LDS per workgroup (K-chunked variant not built yet)",
                     cin, cout, bytes(th));
  if (th > h) th = h;
  a.th = th;
  a.tiles_x = (wdt + a.wc - 1) / a.wc;
  a.tiles_y = (h + th - 1) / th;
  const long long nb = (long long)n * a.tiles_x * a.tiles_y;
  if (nb > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "conv3x3: batch too large for one launch");
  a.nblocks = (unsigned)nb;
  const size_t lds_bytes = bytes(th);
  const bool fullm = (cout % 32 == 0);
  if (cin == 3 && cout == 64) return launch_k<14, 2, true>(a, lds_bytes, s);
  if (cin == 3 && cout <= 64) return launch_k<14, 2, false>(a, lds_bytes, s);
  if (cin == 1 && cout <= 64) return launch_k<5, 2, false>(a, lds_bytes, s);
  return fullm ? launch_k<0, 2, true>(a, lds_bytes, s) : launch_k<0, 2, false>(a, lds_bytes, s);
}

}  // namespace mv
