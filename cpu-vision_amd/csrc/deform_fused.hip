// deform_fused.hip -- torchvision::deform_conv2d forward (csrc/ops/deform_conv2d.cpp:164-169; CPU kernel
// csrc/ops/cpu/deform_conv2d_kernel.cpp) as ONE kernel on gfx950: the deformable im2col columns never exist in HBM.
//
// deform.hip writes columns[b][(c*kh + i)*kw + j][pixel] to a workspace and runs a GEMM over it (0.6 GB written and read back
// for 8 x 256 x 64 x 64, 3x3).  Here a workgroup owns a tile of output pixels of one image (16 x 8, or 16 x 4 with 256 output
// channels) x up to 256 output channels of one weight group and walks K = (channel, ky, kx) in chunks of `cb` channels:
//   window   the input window the tile's samples can reach while |offset| <= kFMargin, cb channels, staged with coalesced
//            loads (rows / columns outside the image are zeros = what bilinear_interpolate, deform_conv2d_kernel.cpp:80-116,
//            substitutes for a corner outside the image); the next chunk's window and weights are in flight in registers
//   params   per (tap, pixel) of the tile: (lh, lw, mask, window offset of the top-left corner) -- they depend on the offset
//            group only, so they are computed once per workgroup (and again when the K walk enters the next offset group)
//   gather   xs[c*taps + tap][pixel] = mask * (w1*v1 + w2*v2 + w3*v3 + w4*v4), the reference's operations in its order, the
//            four corners from LDS; a sample whose 2 x 2 neighbourhood leaves the window takes the global-memory path
//   MFMA     v_mfma_f32_32x32x2_f32 over the chunk: rows = pixels (xs), columns = output channels (W tile), one accumulator
//            per output in ascending k = the oracle's chain (orc_deform_conv2d_f32), bias (+ activation) in the epilogue
// Producer and consumer waves (below).  What the gather costs is paid once per pixel tile, so the tile takes as many output
// channels as the accumulators hold: 256 (4 waves x 2 x 2 tiles of 32 x 32).
#include <cstdlib>

#include "mv_common.h"
#include "mv_epilogue.h"

namespace mv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-byte access
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef MV_DF_ABLATE
// tools/ab_deform.py builds ablation variants: 1 no gather, 2 no MFMAs, 4 no W staging, 16 no window staging,
// 32 MFMA operands read once per chunk, 64 no corner reads
#define MV_DF_ABLATE 0
#endif
#ifndef MV_DF_PRIO
#define MV_DF_PRIO 1  // 0 none, 1 consumer waves first, 2 producer waves first (0.454 / 0.452 / 0.463 ms: profiles/r02_ab_deform_roles_final.log)
#endif
#ifndef MV_DF_DEPTH
#define MV_DF_DEPTH 3
#endif
constexpr int kFTW = 16;                 // tile width in output pixels (a 32-pixel MFMA tile = two rows of the tile)
constexpr int kFMargin = 6;              // |offset| served from the staged window
constexpr int kFWinU = 12;               // window floats per producer thread and chunk (registers)
constexpr int kFWU = 10;                 // W float4 per producer thread and chunk (registers)
constexpr int kFKMax = 40;               // K (cb * taps) per chunk at most
constexpr int kFDepth = MV_DF_DEPTH;     // k-steps of MFMA operands in flight in a consumer wave
constexpr int kFTP = 36;                 // transpose buffer pitch of the epilogue

// tile geometry per MW = 32-channel tiles of a workgroup: pixels, (channel, pixel) tiles per consumer wave
__host__ __device__ constexpr int fused_px(int mw) { return mw == 8 ? 64 : 128; }
__host__ __device__ constexpr int fused_mt(int mw) { return mw == 8 ? 2 : 1; }
__host__ __device__ constexpr int fused_pt(int mw) { return mw == 8 ? 2 : mw; }
// W tile pitch: even (8-byte stores of k pairs) with pitch / 2 odd -- the 32 channel rows a wave reads land on the 32 even
// banks, the k + 1 half of the wave on the odd ones
__host__ __device__ constexpr int fused_w_pitch(int kcp) { return (kcp / 2) % 2 ? kcp : kcp + 2; }

struct DeformFusedArgs {
  const float* x;
  const float* w;
  const float* offset;
  const float* mask;  // null: no modulation
  const float* bias;  // null: none
  float* y;
  int cin, cout, h, wd, kh, kw, sh, sw, ph, pw, dh, dw, oh, ow;
  int groups, offset_groups, act;
  int tiles_x, tiles_y, mblocks;
  int cb, kc, kcp, kq, wp;       // channels per chunk; K per chunk (cb * taps), rounded up to even, in float4, W tile pitch
  int win_h, win_w, win_pitch;   // staged window
  int xs_off, win_off, par_off;  // LDS regions (floats): 2 W tiles at 0, 2 column tiles, 2 windows, the params
  int vec_y;
};

// the reference's sample, corners from global memory (deform_conv2d_kernel.cpp:80-116 with the channel-independent part first)
struct SlowSample {
  float w1, w2, w3, w4;
  int o1, o2, o3, o4;
  bool ok1, ok2, ok3, ok4, outside;
};

__device__ inline SlowSample slow_sample(float h, float w, int H, int W) {
  SlowSample s;
  s.outside = (h <= -1 || H <= h || w <= -1 || W <= w);
  const int h_low = (int)floorf(h), w_low = (int)floorf(w);
  const int h_high = h_low + 1, w_high = w_low + 1;
  const float lh = h - h_low, lw = w - w_low;
  const float hh = 1 - lh, hw = 1 - lw;
  s.ok1 = !s.outside && h_low >= 0 && w_low >= 0;
  s.ok2 = !s.outside && h_low >= 0 && w_high <= W - 1;
  s.ok3 = !s.outside && h_high <= H - 1 && w_low >= 0;
  s.ok4 = !s.outside && h_high <= H - 1 && w_high <= W - 1;
  s.o1 = s.ok1 ? h_low * W + w_low : 0, s.o2 = s.ok2 ? h_low * W + w_high : 0;
  s.o3 = s.ok3 ? h_high * W + w_low : 0, s.o4 = s.ok4 ? h_high * W + w_high : 0;
  s.w1 = hh * hw, s.w2 = hh * lw, s.w3 = lh * hw, s.w4 = lh * lw;
  return s;
}

// MW: 32-channel tiles per workgroup (1, 2, 4: 128 pixels; 8: 64 pixels); TAPS / CB: compile-time kh*kw and channels per chunk
// (0 = run time).
// 512 threads in two roles (wave-uniform): waves 0-3 CONSUME chunk `it` (MFMAs over xs[it & 1], wl[it & 1]) while waves 4-7
// PRODUCE chunk it + 1 (W tile, gather) into the other buffers and stage the window of chunk it + 2 -- one barrier per chunk.
// What was measured on the way (profiles/r02_micro_mfma_*.log, r02_trace_deform_*.log, r02_ab_deform_*.log):
//   * fp32 MFMAs and fp32 VALU work do not overlap on a SIMD, neither from another wave nor in the same wave's instruction
//     stream: 72 MFMAs 1.97 us, + 8 VALU ops per MFMA 2.68 us, + 16 per MFMA 3.29 us (2.7 cycles of matrix time per VALU op);
//     LDS instructions cost the MFMA wave ~7-12 cycles each.  A chunk therefore costs MFMAs + every other instruction, and the
//     only thing the roles buy is that the producer's LATENCIES (global loads, LDS round trips) run under the MFMAs.
//   * while the consumer wave of a SIMD issues MFMAs, the producer wave gets about one instruction in per MFMA and runs at full
//     speed only once the consumer waits at the barrier -- whatever s_setprio says.  So the producer's loop starts with what has
//     a latency (corner reads, loads), uses as few instructions as possible for it (buffer loads with scalar chunk offsets: no
//     address arithmetic; loop-invariant LDS offsets), and ends with the arithmetic.
//   * two independent 256-thread workgroups per CU, each alternating gather and MFMA phases, ran in lockstep: 0.54 ms where
//     the gather alone took 0.30 and the MFMAs alone 0.30.
// 8 x 256 x 64 x 64 -> 256: columns workspace + GEMM 0.61 ms; this kernel with 128-channel tiles 0.51, with 256-channel tiles
// 0.45 ms = 85 TFLOP/s (MFMA time alone: 0.25 ms).
template <int MW, int TAPS, int CB>
__global__ __launch_bounds__(512, 2) void k_deform_fused(const DeformFusedArgs A) {
  constexpr int PX = fused_px(MW), XP = PX | 32;  // xs pitch: the two k rows of an MFMA step fall in different bank halves
  constexpr int TH = PX / kFTW;
  constexpr int MT = fused_mt(MW), PT = fused_pt(MW);
  constexpr int R = MW * 32;  // W rows
  constexpr int KCT = TAPS * CB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid_all = threadIdx.x;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid_all >> 6);
  const bool producer = wave_all >= 4;  // wave-uniform role
  const int tid = tid_all & 255, lane = tid & (kWave - 1), wave = wave_all & 3;
#ifdef MV_DF_TRACE
  // tools/trace_deform.py: cycle stamps of block 0's first producer and consumer wave over chunks 10..13, written behind y
  // (the script allocates the extra room; shapes whose tiles cover the output exactly)
  int trace_it = -1;
  auto stamp = [&](int slot) {
    if (blockIdx.x == 0 && (tid_all & 63) == 0 && (wave_all & 3) == 0 && trace_it >= 10 && trace_it < 14)
      reinterpret_cast<long long*>(A.y + (size_t)gridDim.x / A.mblocks * PX * A.cout)[(producer ? 256 : 0) + (trace_it - 10) * 16 + slot] =
          __builtin_readcyclecounter();
  };
#define MV_DF_STAMP(it, slot) (trace_it = (it), stamp(slot))
#else
#define MV_DF_STAMP(it, slot) ((void)0)
#endif
  const int taps = TAPS ? TAPS : A.kh * A.kw;
  const int cb = CB ? CB : A.cb;
  const int KC = KCT ? KCT : A.kc, KCP = KCT ? ((KCT + 1) & ~1) : A.kcp, KQ = KCT ? (KCT + 3) / 4 : A.kq;
  const int WP = KCT ? fused_w_pitch(KCP) : A.wp;
  const int H = A.h, W = A.wd;
  float* const wl0 = lds;                                          // 2 x ([R][WP] + a spare slot)
  float* const xs0 = lds + A.xs_off;                               // 2 x [KCP][XP]; the epilogue's transpose buffers
  float* const win0 = lds + A.win_off;                             // 2 x ([win_h][win_pitch][cb] + a spare slot)
  f32x4* const par = reinterpret_cast<f32x4*>(lds + A.par_off);    // [taps][PX]
  const int wl_sz = A.xs_off / 2, xs_sz = (A.win_off - A.xs_off) / 2, win_sz = (A.par_off - A.win_off) / 2;

  auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };  // workgroup-uniform: keep it in a scalar register
  unsigned bid = blockIdx.x;
  const int mb = uni(bid % A.mblocks);
  bid /= A.mblocks;
  const int g = uni(bid % A.groups);
  bid /= A.groups;
  const int tile_x = uni(bid % A.tiles_x);
  bid /= A.tiles_x;
  const int tile_y = uni(bid % A.tiles_y);
  const int b = uni(bid / A.tiles_y);
  const int cg = uni(A.cin / A.groups), mg = uni(A.cout / A.groups), cog = uni(A.cin / A.offset_groups);
  const int j0 = g * mg + mb * R, jend = (g + 1) * mg;
  const int oy0 = tile_y * TH, ox0 = tile_x * kFTW;
  const int chunks = uni(cg / cb);

  if (producer) {
    const int Kg = cg * taps;  // a weight row
    const int wy0 = oy0 * A.sh - A.ph - kFMargin, wx0 = ox0 * A.sw - A.pw - kFMargin;  // window origin (may be negative)
    const int wh = A.win_h, ww = A.win_w, wp = A.win_pitch;
    const size_t ohw = (size_t)A.oh * A.ow;
    const float* xg = A.x + ((size_t)b * A.cin + (size_t)g * cg) * H * W;  // the weight group's first channel

    // ---- this thread's window elements: the same positions of every chunk.  The window is CHANNEL-INTERLEAVED,
    //      win[(ly * wp + lx) * cb + c]: a sample's corner is then ONE 16-byte LDS read for the 4 channels of a chunk instead of
    //      a 4-byte read per channel -- scattered 16-byte reads collide on far fewer banks than scattered 4-byte reads (the gather
    //      was LDS-bound at 4 x the conflict-free time).  Element u of a thread = channel u % cb of its position u / cb.
    //      Straight-line staging: the buffers are zeroed once; a position outside the image (or past the thread's share) loads
    //      a valid address and stores into the buffer's spare slot, so that its zeros stay
    // byte offsets from the chunk's first channel for buffer loads (resource + scalar chunk offset + this): no address arithmetic
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xg), 0, -1, 0x00020000);
    unsigned wsrc[kFWinU];
    int wdst[kFWinU];
    {
#pragma unroll
      for (int u = 0; u < kFWinU; ++u) {
        const int i = tid + 256 * (u / cb), c = u % cb;
        wsrc[u] = 0, wdst[u] = win_sz - 4 + (c & 3);
        if (i < wh * ww && u < kFWinU / cb * cb) {
          const int ly = i / ww, lx = i - ly * ww;
          const int gy = wy0 + ly, gx = wx0 + lx;
          if (gy >= 0 && gy < H && gx >= 0 && gx < W) wsrc[u] = 4u * ((c * H + gy) * W + gx), wdst[u] = (ly * wp + lx) * cb + c;
        }
      }
    }
    for (int i = tid; i < 2 * win_sz; i += 256) win0[i] = 0.f;  // positions outside the image: zeros, never stored to
    float winreg[kFWinU];
    auto gload_window = [&](int it) {
      const int chunk_bytes = uni(it * cb * H * W * 4);  // scalar offset of the chunk's first channel
#pragma unroll
      for (int u = 0; u < kFWinU; ++u) winreg[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, wsrc[u], chunk_bytes, 0));
    };
    auto store_window = [&](int buf) {
      float* win = win0 + buf * win_sz;
      if (CB == 4) {
#pragma unroll
        for (int u = 0; u < kFWinU; u += 4)
          *reinterpret_cast<f32x4*>(win + wdst[u]) = (f32x4){winreg[u], winreg[u + 1], winreg[u + 2], winreg[u + 3]};
      } else {
#pragma unroll
        for (int u = 0; u < kFWinU; ++u) win[wdst[u]] = winreg[u];
      }
    };
    // ---- W tile.  Rows past the weight group's last channel read its last row: their outputs are never stored.
    //      Whole float4 chunks (KC % 4 == 0, the 3x3 case): float4 number idx = tid + 256 u of the tile in row-major order, so
    //      that a wave reads a few contiguous rows; LDS and global offsets are loop-invariant registers, the surplus of the
    //      last round goes to the tile's spare slot.  Otherwise: thread = (row, part), element-wise tails
    constexpr bool WFLAT = KCT && KCT % 4 == 0;
    constexpr int KQT = WFLAT ? KCT / 4 : 1;
    constexpr int TPR = 256 / R;
    constexpr int WU = WFLAT ? (R * KQT + 255) / 256 : (kFKMax / 4 + TPR - 1) / TPR;
    static_assert(WU <= kFWU, "W registers");
    unsigned wgo[WU];  // byte offsets from the tile's first row
    int wlo[WU];
    if (WFLAT) {
#pragma unroll
      for (int u = 0; u < WU; ++u) {
        const int idx = tid + 256 * u, row = min(idx / KQT, R - 1), q = idx % KQT;
        wgo[u] = 4u * ((min(j0 + row, jend - 1) - j0) * Kg + 4 * q);
        wlo[u] = idx < R * KQT ? row * WP + 4 * q : wl_sz - 4;
      }
    }
    const int wrow = tid % R, wpart = tid / R;
    const float* wsrc_row = A.w + (size_t)(WFLAT ? j0 : min(j0 + wrow, jend - 1)) * Kg;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wsrc_row), 0, -1, 0x00020000);
    f32x4 wreg[WU];
    auto gload_weights = [&](int it) {
      const float* src = wsrc_row + (size_t)it * KC;
#pragma unroll
      for (int u = 0; u < WU; ++u) {
        if (WFLAT) {
          wreg[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wgo[u], uni(it * KC * 4), 0));
        } else {  // the last float4 of a chunk may end past it (and past the array): element-wise, zeros for k >= KC
          const int q = min(wpart + TPR * u, KQ - 1);
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (4 * q + 3 < KC) {
            v = *reinterpret_cast<const f32x4u*>(src + 4 * q);
          } else {
            if (4 * q + 0 < KC) v.x = src[4 * q];
            if (4 * q + 1 < KC) v.y = src[4 * q + 1];
            if (4 * q + 2 < KC) v.z = src[4 * q + 2];
          }
          wreg[u] = v;
        }
      }
    };
    auto store_weights = [&](int buf) {  // 8-byte stores: WP is even (and WP / 2 odd: the consumers' reads spread over all banks)
      float* wl = wl0 + buf * wl_sz;
#pragma unroll
      for (int u = 0; u < WU; ++u) {
        if (WFLAT) {
          *reinterpret_cast<f32x2*>(wl + wlo[u]) = (f32x2){wreg[u].x, wreg[u].y};
          *reinterpret_cast<f32x2*>(wl + wlo[u] + 2) = (f32x2){wreg[u].z, wreg[u].w};
        } else {
          const int q = wpart + TPR * u;
          if (4 * q < KCP) *reinterpret_cast<f32x2*>(wl + wrow * WP + 4 * q) = (f32x2){wreg[u].x, wreg[u].y};
          if (4 * q + 2 < KCP) *reinterpret_cast<f32x2*>(wl + wrow * WP + 4 * q + 2) = (f32x2){wreg[u].z, wreg[u].w};
        }
      }
    };

    // ---- per (tap, pixel): the channel-independent part of the sample (written and read by the same thread)
    constexpr int NPART = 256 / PX;  // producer threads per pixel: thread `part` takes taps part, part + NPART, ...
    const int pt = tid % PX, part = tid / PX;
    const int my_oy = oy0 + pt / kFTW, my_ox = ox0 + pt % kFTW;
    const bool my_live = my_oy < A.oh && my_ox < A.ow;
    const size_t my_pix = (size_t)min(my_oy, A.oh - 1) * A.ow + min(my_ox, A.ow - 1);
    auto sample_pos = [&](int og, int mi, float& h, float& w, float& mv) {
      const int i = mi / A.kw, j = mi - i * A.kw;
      const float* offp = A.offset + ((size_t)b * A.offset_groups + og) * 2 * taps * ohw + my_pix;
      mv = A.mask ? A.mask[(((size_t)b * A.offset_groups + og) * taps + mi) * ohw + my_pix] : 1.f;
      h = (my_oy * A.sh - A.ph + i * A.dh) + offp[(size_t)(2 * mi) * ohw];
      w = (my_ox * A.sw - A.pw + j * A.dw) + offp[(size_t)(2 * mi + 1) * ohw];
    };
    auto compute_params = [&](int og) {
      for (int mi = part; mi < taps; mi += NPART) {
        // a pixel outside the image: mask 0 x the window's first element (its outputs are never stored)
        f32x4 P = {0.f, 0.f, 0.f, __int_as_float(0)};
        if (my_live) {
          float h, w, mv;
          sample_pos(og, mi, h, w, mv);
          const int h_low = (int)floorf(h), w_low = (int)floorf(w);
          const int ly = h_low - wy0, lx = w_low - wx0;
          // The window serves samples INSIDE the image only: for h <= -1, H <= h, w <= -1 or W <= w the reference returns exactly
          // 0 (deform_conv2d_kernel.cpp:88-90) where the window would compute 0 x (the border pixel) -- NaN if that pixel is not
          // finite.  Those samples, huge offsets and NaN positions (every comparison false) go to gather_far, which applies the
          // reference's `outside` test.
          const bool inside = h > -1 && h < A.h && w > -1 && w < A.wd;
          const bool inwin = inside && ly >= 0 && ly + 1 < wh && lx >= 0 && lx + 1 < ww;
          P.x = h - h_low, P.y = w - w_low, P.z = mv;
          P.w = __int_as_float(inwin ? ly * wp + lx : -1);
        }
        par[mi * PX + pt] = P;
      }
    };
    // one (tap, pixel) x the chunk's channels, corners from the window.  Branch-free for every lane (a lane whose sample left
    // the window reads offset 0 and is overwritten by gather_far).  Reads and writes are separate steps: the compiler keeps
    // LDS writes and reads in program order (it cannot tell the column tile from the window), so interleaved they would pay
    // one LDS round trip per value
    struct TapRead {
      f32x4 P;
      float v[4][CB ? CB : 8];  // [corner][channel]
    };
    auto tap_read = [&](const float* win, int mi, TapRead& r) {
      r.P = par[mi * PX + pt];
      const float* q = win + max(__float_as_int(r.P.w), 0) * cb;
#if MV_DF_ABLATE & 64
      if (CB == 4) {
#pragma unroll
        for (int c = 0; c < 4; ++c) r.v[0][c] = r.P.x, r.v[1][c] = r.P.y, r.v[2][c] = r.P.z, r.v[3][c] = r.P.x + c;
        return;
      }
#endif
      if (CB == 4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(q), bq = *reinterpret_cast<const f32x4*>(q + 4);
        const f32x4 c2 = *reinterpret_cast<const f32x4*>(q + wp * 4), d = *reinterpret_cast<const f32x4*>(q + wp * 4 + 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) r.v[0][c] = a[c], r.v[1][c] = bq[c], r.v[2][c] = c2[c], r.v[3][c] = d[c];
      } else {
#pragma unroll
        for (int c = 0; c < (CB ? CB : 8); ++c) {
          if (CB == 0 && c >= cb) break;
          r.v[0][c] = q[c], r.v[1][c] = q[cb + c], r.v[2][c] = q[wp * cb + c], r.v[3][c] = q[(wp + 1) * cb + c];
        }
      }
    };
    auto tap_write = [&](float* xs, int mi, const TapRead& r) {
      const float lh = r.P.x, lw = r.P.y, mv = r.P.z;
      const float hh = 1 - lh, hw = 1 - lw;
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
      float* dst = xs + mi * XP + pt;
      const int crow = taps * XP;  // xs rows between consecutive channels
#pragma unroll
      for (int c = 0; c < (CB ? CB : 8); ++c) {
        if (CB == 0 && c >= cb) break;
        float val = w1 * r.v[0][c];
        val = val + w2 * r.v[1][c];
        val = val + w3 * r.v[2][c];
        val = val + w4 * r.v[3][c];
        dst[c * crow] = mv * val;
      }
      return __float_as_int(r.P.w) < 0;
    };
    auto gather_far = [&](float* xs, int it, int og, int mi) {  // the reference's sample from global memory
      if (__float_as_int(par[mi * PX + pt].w) >= 0) return;
      float h, w, mv;
      sample_pos(og, mi, h, w, mv);
      const SlowSample s = slow_sample(h, w, H, W);
      const float* in = xg + (size_t)it * cb * H * W;
      float* dst = xs + mi * XP + pt;
      for (int c = 0; c < cb; ++c, in += (size_t)H * W) {
        const float v1 = s.ok1 ? in[s.o1] : 0.f, v2 = s.ok2 ? in[s.o2] : 0.f, v3 = s.ok3 ? in[s.o3] : 0.f, v4 = s.ok4 ? in[s.o4] : 0.f;
        float val = s.w1 * v1;
        val = val + s.w2 * v2;
        val = val + s.w3 * v3;
        val = val + s.w4 * v4;
        dst[c * taps * XP] = mv * (s.outside ? 0.f : val);
      }
    };
    // the gather of a chunk in two steps, so that the main loop can put the other staging work between the corner reads and the
    // arithmetic (TAPS known: all of the thread's taps are read at once)
    constexpr int NTAP = TAPS ? (TAPS + NPART - 1) / NPART : 1;
    TapRead tr[NTAP];
    auto gather_reads = [&](int it) {
      if (!TAPS) return;
      const float* win = win0 + (it & 1) * win_sz;
#pragma unroll
      for (int u = 0; u < NTAP; ++u) tap_read(win, min(NPART * u + part, (TAPS ? TAPS : 1) - 1), tr[u]);
    };
    auto gather_writes = [&](int it, int og) {
      const float* win = win0 + (it & 1) * win_sz;
      float* xs = xs0 + (it & 1) * xs_sz;
      bool far = false;
      if (TAPS) {
#pragma unroll
        for (int u = 0; u < NTAP; ++u)
          if (NPART * u + part < TAPS) far |= tap_write(xs, NPART * u + part, tr[u]);
      } else {
        for (int mi = part; mi < taps; mi += NPART) {
          TapRead r;
          tap_read(win, mi, r);
          far |= tap_write(xs, mi, r);
        }
      }
      if (far)
        for (int mi = part; mi < taps; mi += NPART) gather_far(xs, it, og, mi);
    };
    auto gather = [&](int it, int og) {
      gather_reads(it);
      __builtin_amdgcn_sched_barrier(0);
      gather_writes(it, og);
    };
    // a chunk's offset group (cb divides cog): og_left chunks remain in the current one
    const int chunks_per_og = uni(cog / cb);
    int cur_og = uni((g * cg) / cog), og_left = chunks_per_og - uni(((g * cg) % cog) / cb);

    if (KCP > KC && tid < PX) xs0[KC * XP + tid] = 0.f, xs0[xs_sz + KC * XP + tid] = 0.f;  // pad row of an odd chunk
#if MV_DF_PRIO == 2
    __builtin_amdgcn_s_setprio(2);
#endif
    compute_params(cur_og);
    gload_window(0);
    gload_weights(0);
    __syncthreads();  // the windows are zeroed
    store_window(0);
    store_weights(0);
    __syncthreads();  // window 0 visible to every producer
    if (chunks > 1) gload_window(1), gload_weights(1);
    gather(0, cur_og);
    if (chunks > 1) store_window(1);
    __syncthreads();  // chunk 0 ready
    for (int it = 0; it < chunks; ++it) {
      // consumers: chunk it.  Here: chunk it + 1 into the other buffers (free since the barrier), window it + 2
      MV_DF_STAMP(it, 0);
      // While the consumer wave of this SIMD issues MFMAs, this wave gets about one instruction in per MFMA and runs at full speed
      // only once the consumer waits at the barrier (profiles/r02_trace_deform_roles.log).  So: first what has a latency to hide
      // (corner reads, global loads), with as few instructions as possible; the arithmetic last
      if (it + 1 < chunks) {
        if (--og_left == 0) og_left = chunks_per_og, compute_params(++cur_og);
        const int og = cur_og;
#if !(MV_DF_ABLATE & 1)
        gather_reads(it + 1);
        __builtin_amdgcn_sched_barrier(0);
#endif
        MV_DF_STAMP(it, 1);
#if !(MV_DF_ABLATE & 16)
        if (it + 2 < chunks) gload_window(it + 2);
        __builtin_amdgcn_sched_barrier(0);
#endif
#if !(MV_DF_ABLATE & 4)
        store_weights((it + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (it + 2 < chunks) gload_weights(it + 2);
        __builtin_amdgcn_sched_barrier(0);
#endif
        MV_DF_STAMP(it, 2);
#if !(MV_DF_ABLATE & 16)
        // (its buffer was last read by gather(it), before the barrier that opened this iteration; the corner reads above use the
        // other one)
        if (it + 2 < chunks) store_window(it & 1);
        __builtin_amdgcn_sched_barrier(0);
#endif
        MV_DF_STAMP(it, 5);
#if !(MV_DF_ABLATE & 1)
        gather_writes(it + 1, og);
#endif
        MV_DF_STAMP(it, 6);
      }
      __syncthreads();
      MV_DF_STAMP(it, 7);
    }
    return;
  }

  // ---- consumers: wave = MT channel tiles x PT pixel tiles
  const int l31 = lane & 31, hf = lane >> 5;
  const int mt0 = MW == 8 ? 2 * wave : wave % MW, pg = MW == 8 ? 0 : wave / MW;
  const bool live = j0 + mt0 * 32 < jend;  // wave-uniform
  f32x16 acc[MT][PT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < PT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][t][i] = 0.f;
  __syncthreads();
  __syncthreads();
  __syncthreads();  // chunk 0 ready
#if MV_DF_PRIO == 1
  __builtin_amdgcn_s_setprio(2);
#endif
  for (int it = 0; it < chunks; ++it) {
    MV_DF_STAMP(it, 0);
#if MV_DF_ABLATE & 2
    if (A.act == 77) {
#else
    if (live) {
#endif
      const float* wpnt = wl0 + (it & 1) * wl_sz + (mt0 * 32 + l31) * WP + hf;        // W[channel l31 of tile m][2s + hf]
      const float* xpnt = xs0 + (it & 1) * xs_sz + hf * XP + pg * (PT * 32) + l31;    // columns[2s + hf][pixel l31 of tile t]
      if (KCT) {
        // straight-line: the operands of k-step s + kFDepth are read before the MFMAs of k-step s issue (the producers'
        // scattered reads share the LDS queue)
        constexpr int KS = KCT ? (KCT + 1) / 2 : 1, D = kFDepth < KS ? kFDepth : KS;
        float wq[D][MT], xq[D][PT];
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
          for (int m = 0; m < MT; ++m) wq[d][m] = wpnt[m * 32 * WP + 2 * d];
#pragma unroll
          for (int t = 0; t < PT; ++t) xq[d][t] = xpnt[2 * d * XP + t * 32];
        }
#if MV_DF_ABLATE & 32
#define MV_DF_OPERANDS(s) false
#else
#define MV_DF_OPERANDS(s) ((s) + D < KS)
#endif
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          float wc[MT], xc[PT];
#pragma unroll
          for (int m = 0; m < MT; ++m) wc[m] = wq[s % D][m];
#pragma unroll
          for (int t = 0; t < PT; ++t) xc[t] = xq[s % D][t];
          if (MV_DF_OPERANDS(s)) {
#pragma unroll
            for (int m = 0; m < MT; ++m) wq[s % D][m] = wpnt[m * 32 * WP + 2 * (s + D)];
#pragma unroll
            for (int t = 0; t < PT; ++t) xq[s % D][t] = xpnt[2 * (s + D) * XP + t * 32];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < PT; ++t) acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xc[t], wc[m], acc[m][t], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        for (int s = 0; s < KCP / 2; ++s) {
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float wv = wpnt[m * 32 * WP + 2 * s];
#pragma unroll
            for (int t = 0; t < PT; ++t)
              acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(xpnt[2 * s * XP + t * 32], wv, acc[m][t], 0, 0, 0);
          }
        }
      }
    }
    MV_DF_STAMP(it, 1);
    __syncthreads();
    MV_DF_STAMP(it, 2);
  }

  // ---- epilogue: each 32 x 32 tile through a wave-private [channel][pixel] buffer, so that a lane stores 4 consecutive pixels
  //      of one channel (16 B); a tile of the accumulators is pixels 32t .. 32t+31 of the 16-wide tile = two rows of 16.
  //      (The barrier that closed the last chunk ended every operand read; the producers are gone.)
  if (!live) return;
  float* tb = xs0 + wave * (32 * kFTP);
  const int r0 = lane >> 3, q = (lane & 7) * 4;
  const Epilogue E = {A.bias, nullptr, nullptr, nullptr, 0, A.act};
  const Clamp cl = make_clamp(A.act);
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) {
    const int mbase = j0 + (mt0 + mi) * 32;
    ChannelTerms ct[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ct[j] = channel_terms(E, min(mbase + r0 + 8 * j, A.cout - 1));
#pragma unroll
    for (int t = 0; t < PT; ++t) {
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<f32x4*>(tb + l31 * kFTP + 8 * gq + 4 * hf) =
            (f32x4){acc[mi][t][4 * gq], acc[mi][t][4 * gq + 1], acc[mi][t][4 * gq + 2], acc[mi][t][4 * gq + 3]};
      const int p = pg * (PT * 32) + 32 * t + q;  // pixel of the tile
      const int oy = oy0 + p / kFTW, ox = ox0 + p % kFTW;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = mbase + r0 + 8 * j;
        const f32x4 a = *reinterpret_cast<const f32x4*>(tb + (r0 + 8 * j) * kFTP + q);
        if (m < jend && oy < A.oh && ox < A.ow) {
          float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = epi_norm(v[i], ct[j], E);
            v[i] = (A.act == 3) ? epi_act<1>(v[i], cl) : ((A.act == 4) ? epi_act<2>(v[i], cl) : epi_act<0>(v[i], cl));
          }
          float* dst = A.y + (((size_t)b * A.cout + m) * A.oh + oy) * A.ow + ox;
          if (A.vec_y && ox + 3 < A.ow) {
            *reinterpret_cast<f32x4*>(dst) = (f32x4){v[0], v[1], v[2], v[3]};
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (ox + i < A.ow) dst[i] = v[i];
          }
        }
      }
    }
  }
}

// ---- geometry -> tiles: one place for the launcher and for deform_fused_supported()
struct FusedPlan {
  bool ok;
  int mw, cb, kc, kcp, kq, wp, win_h, win_w, win_pitch, xs_off, win_off, par_off;
  size_t lds_bytes;
};

static int gcd_int(int a, int b) {
  while (b) {
    const int t = a % b;
    a = b, b = t;
  }
  return a;
}

static FusedPlan fused_plan(int cin, int cout, int h, int wd, int kh, int kw, int sh, int sw, int dh, int dw, int groups,
                            int offset_groups) {
  FusedPlan p = {};
  const int taps = kh * kw, cg = cin / groups, mg = cout / groups, cog = cin / offset_groups;
  p.mw = mg <= 32 ? 1 : (mg <= 64 ? 2 : (mg <= 128 ? 4 : 8));
  if (const char* e = tune_env("MV_DF_MW")) p.mw = (atoi(e) == 4 && p.mw == 8) ? 4 : p.mw;
  const int px = fused_px(p.mw), th = px / kFTW;
  p.win_h = (th - 1) * sh + (kh - 1) * dh + 2 + 2 * kFMargin;
  p.win_w = (kFTW - 1) * sw + (kw - 1) * dw + 2 + 2 * kFMargin;
  p.win_pitch = p.win_w | 1;
  if (taps > kFKMax || (long long)p.win_h * p.win_w > kFWinU * 256) return p;
  const int common = gcd_int(cg, cog);  // a chunk stays inside one weight group and one offset group
  int cb = 8;
  const int positions = (p.win_h * p.win_w + 255) / 256;  // window positions per producer thread, x cb channels each
  while (cb > 1 && (common % cb || cb * taps > kFKMax || cb * positions > kFWinU)) cb /= 2;
  if ((long long)cg * h * wd >= 0x1fffffffLL || (long long)p.mw * 32 * cg * taps >= 0x1fffffffLL) return p;  // byte offsets stay below 2^31
  p.cb = cb;
  p.kc = cb * taps, p.kcp = (p.kc + 1) & ~1, p.kq = (p.kc + 3) / 4, p.wp = fused_w_pitch(p.kcp);
  const int wl = ((p.mw * 32 * p.wp + 3) & ~3) + 4;                                              // x 2 buffers, + a spare slot
  const int xs = p.kcp * (px | 32) > 4 * 32 * kFTP / 2 ? p.kcp * (px | 32) : 4 * 32 * kFTP / 2;  // x 2 (>= the transpose buffers)
  const int win = ((cb * p.win_h * p.win_pitch + 3) & ~3) + 4;                                  // x 2, + the spare slot
  p.xs_off = 2 * wl;
  p.win_off = p.xs_off + 2 * xs;
  p.par_off = p.win_off + 2 * win;
  p.lds_bytes = sizeof(float) * ((size_t)p.par_off + 4 * (size_t)taps * px);
  p.ok = p.lds_bytes <= 160 * 1024;  // one 512-thread workgroup per CU
  return p;
}

bool deform_fused_supported(int cin, int cout, int h, int wd, int kh, int kw, int sh, int sw, int dh, int dw, int groups,
                            int offset_groups) {
  return fused_plan(cin, cout, h, wd, kh, kw, sh, sw, dh, dw, groups, offset_groups).ok;
}

// workgroups of the fused launch (0: unsupported).  One workgroup per CU: below 256 the two-kernel form, which cuts the GEMM into
// smaller tiles, fills the chip better (4 x 512 x 32 x 32 -> 512: 128 workgroups, 0.44 ms fused, 0.30 ms with the workspace)
int64_t deform_fused_workgroups(int64_t n, int cin, int cout, int h, int wd, int kh, int kw, int sh, int sw, int ph, int pw, int dh,
                                int dw, int groups, int offset_groups) {
  const FusedPlan p = fused_plan(cin, cout, h, wd, kh, kw, sh, sw, dh, dw, groups, offset_groups);
  if (!p.ok) return 0;
  const int oh = (h + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1, ow = (wd + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  const int th = fused_px(p.mw) / kFTW;
  return n * ((oh + th - 1) / th) * ((ow + kFTW - 1) / kFTW) * groups * ((cout / groups + p.mw * 32 - 1) / (p.mw * 32));
}

template <int MW>
static int fused_launch(const DeformFusedArgs& a, const FusedPlan& p, long long blocks, hipStream_t s) {
  const int taps = a.kh * a.kw;
  if (taps == 9 && a.cb == 4) {
    if (p.lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_deform_fused<MW, 9, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
    hipLaunchKernelGGL((k_deform_fused<MW, 9, 4>), dim3((unsigned)blocks), dim3(512), p.lds_bytes, s, a);
    return check_launchf("k_deform_fused<%d,3x3,cb4>", MW);
  }
  if (p.lds_bytes > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_deform_fused<MW, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
  hipLaunchKernelGGL((k_deform_fused<MW, 0, 0>), dim3((unsigned)blocks), dim3(512), p.lds_bytes, s, a);
  return check_launchf("k_deform_fused<%d,taps%d,cb%d>", MW, taps, a.cb);
}

int launch_deform_fused(const float* x, const float* weight, const float* offset, const float* mask, const float* bias, float* y,
                        int64_t n, int cin, int h, int wd, int cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                        int groups, int offset_groups, int use_mask, hipStream_t s, int act) {
  const FusedPlan p = fused_plan(cin, cout, h, wd, kh, kw, sh, sw, dh, dw, groups, offset_groups);
  if (!p.ok) return set_error(MV_ERR_UNSUPPORTED, "deform_conv2d: geometry outside the fused kernel's tiles");
  DeformFusedArgs a = {};
  a.x = x, a.w = weight, a.offset = offset, a.mask = use_mask ? mask : nullptr, a.bias = bias, a.y = y;
  a.cin = cin, a.cout = cout, a.h = h, a.wd = wd, a.kh = kh, a.kw = kw, a.sh = sh, a.sw = sw, a.ph = ph, a.pw = pw, a.dh = dh, a.dw = dw;
  a.oh = (h + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  a.ow = (wd + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  a.groups = groups, a.offset_groups = offset_groups, a.act = act;
  const int th = fused_px(p.mw) / kFTW;
  a.tiles_x = (a.ow + kFTW - 1) / kFTW, a.tiles_y = (a.oh + th - 1) / th;
  a.mblocks = (cout / groups + p.mw * 32 - 1) / (p.mw * 32);
  a.cb = p.cb, a.kc = p.kc, a.kcp = p.kcp, a.kq = p.kq, a.wp = p.wp;
  a.win_h = p.win_h, a.win_w = p.win_w, a.win_pitch = p.win_pitch;
  a.xs_off = p.xs_off, a.win_off = p.win_off, a.par_off = p.par_off;
  a.vec_y = (a.ow % 4 == 0) && ((uintptr_t)y % 16 == 0);
  const long long blocks = (long long)n * a.tiles_y * a.tiles_x * groups * a.mblocks;
  if (blocks > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "deform_conv2d: batch too large for one launch");
  if (blocks == 0) return MV_OK;
  if (p.mw == 1) return fused_launch<1>(a, p, blocks, s);
  if (p.mw == 2) return fused_launch<2>(a, p, blocks, s);
  if (p.mw == 4) return fused_launch<4>(a, p, blocks, s);
  return fused_launch<8>(a, p, blocks, s);
}

}  // namespace mv
