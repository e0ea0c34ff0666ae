// conv3x3_mfma.hip -- nn.Conv2d(cin, cout, 3, padding=1) [+bias] [+ReLU] as an implicit GEMM on the
// gfx950 fp32 matrix core.  Replaces `conv2d = nn.Conv2d(in_channels, v, kernel_size=3, padding=1)`
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

namespace mv {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* x;
  const float* w;
  const float* b;
  float* y;
  int cin, cout, h, wdt;
  int K;        // cin * 9
  int ksteps;   // ceil(K / 2)
  int mtiles;   // ceil(cout / 32)
  int th;       // output rows per workgroup
  int wc;       // output columns per workgroup (multiple of 32, <= 256)
  int ntx;      // wc / 32
  int pitch;    // LDS row pitch of the input tile (wc + 2, padded)
  int tiles_x, tiles_y;
  int relu;
  unsigned nblocks;
};

__device__ inline float bias_act(float v, float bias, int relu) {
  v = v + bias;
  if (relu) v = (v > 0.f || v != v) ? v : 0.f;  // torch.relu keeps NaN
  return v;
}

// KS > 0: static k-step count, weights register-resident (requires mtiles <= MT).
// KS == 0: runtime k-steps, weights re-read from LDS per step, any number of 32-channel tiles.
template <int KS, int MT>
__global__ __launch_bounds__(256) void k_conv3x3(const ConvArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
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
  for (int i = tid; i < A.mtiles * ksteps * 64; i += 256) {
    const int l = i & 63, s = (i >> 6) % ksteps, m = (i >> 6) / ksteps;
    const int co = 32 * m + (l & 31), k = 2 * s + (l >> 5);
    wfr[i] = (co < cout && k < K) ? A.w[(size_t)co * K + k] : 0.f;
  }
  // ---- tap k = (ci, dy, dx) -> offset inside the input tile
  for (int k = tid; k < 2 * ksteps; k += 256) {
    const int kk = k < K ? k : 0;
    const int ci = kk / 9, r = kk - 9 * ci, dy = r / 3, dx = r - 3 * dy;
    koff[k] = (ci * tile_rows + dy) * pitch + dx;
  }
  // ---- zero-padded input tile: rows yb-1 .. yb+th, columns xb-1 .. xb+wc
  {
    const int cols = wc + 2;
    const int total = cin * tile_rows * cols;
    for (int i = tid; i < total; i += 256) {
      const int c = i % cols, rr = (i / cols) % tile_rows, ci = i / (cols * tile_rows);
      const int gy = yb - 1 + rr, gx = xb - 1 + c;
      float v = 0.f;
      if (gy >= 0 && gy < h && gx >= 0 && gx < w) v = xp[((size_t)ci * h + gy) * w + gx];
      xin[(ci * tile_rows + rr) * pitch + c] = v;
    }
  }
  __syncthreads();

  const int ntiles = th * A.ntx;
  const size_t plane = (size_t)h * w;

  if constexpr (KS > 0) {
    float afr[MT][KS];
    int off[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      off[s] = koff[2 * s + hf];
#pragma unroll
      for (int m = 0; m < MT; ++m) afr[m][s] = (m < A.mtiles) ? wfr[(m * KS + s) * 64 + lane] : 0.f;
    }
    const bool last_valid = (2 * (KS - 1) + hf) < K;  // K odd: the upper half of the last step is padding
    for (int t = wave; t < ntiles; t += 4) {
      const int ly = t / A.ntx, nx = t - ly * A.ntx;
      const int oy = yb + ly;
      if (oy >= h) break;  // wave-uniform; later tiles of this wave are further down
      const float* bp = xin + ly * pitch + nx * 32 + l31;
      f32x16 acc[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        float bv = bp[off[s]];
        if (s == KS - 1) bv = last_valid ? bv : 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[m][s], bv, acc[m], 0, 0, 0);
      }
      const int px = xb + nx * 32 + l31;
      if (px < w) {
        float* yo = A.y + (size_t)img * cout * plane + (size_t)oy * w + px;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int co = 32 * m + (i & 3) + 8 * (i >> 2) + 4 * hf;
            if (co < cout) {
              const float bias = A.b ? A.b[co] : 0.f;
              __builtin_nontemporal_store(bias_act(acc[m][i], bias, A.relu), yo + (size_t)co * plane);
            }
          }
      }
    }
  } else {
    for (int t = wave; t < ntiles; t += 4) {
      const int ly = t / A.ntx, nx = t - ly * A.ntx;
      const int oy = yb + ly;
      if (oy >= h) break;
      const float* bp = xin + ly * pitch + nx * 32 + l31;
      const int px = xb + nx * 32 + l31;
      for (int m0 = 0; m0 < A.mtiles; m0 += MT) {
        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
        for (int s = 0; s < ksteps; ++s) {
          const int k = 2 * s + hf;
          float bv = bp[koff[k]];
          bv = (k < K) ? bv : 0.f;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const float av = (m0 + m < A.mtiles) ? wfr[((m0 + m) * ksteps + s) * 64 + lane] : 0.f;
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
          }
        }
        if (px < w) {
          float* yo = A.y + (size_t)img * cout * plane + (size_t)oy * w + px;
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const int co = 32 * (m0 + m) + (i & 3) + 8 * (i >> 2) + 4 * hf;
              if (co < cout) {
                const float bias = A.b ? A.b[co] : 0.f;
                __builtin_nontemporal_store(bias_act(acc[m][i], bias, A.relu), yo + (size_t)co * plane);
              }
            }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <int KS, int MT>
static int launch_k(const ConvArgs& a, size_t lds_bytes, hipStream_t s) {
  auto k = k_conv3x3<KS, MT>;
  if (lds_bytes > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
  hipLaunchKernelGGL(k, dim3(a.nblocks), dim3(256), lds_bytes, s, a);
  return check_launch("k_conv3x3");
}

int launch_conv3x3(const float* x, const float* w, const float* b, float* y, int64_t n, int cin, int h, int wdt,
                   int cout, int relu, hipStream_t s) {
  ConvArgs a = {};
  a.x = x, a.w = w, a.b = b, a.y = y;
  a.cin = cin, a.cout = cout, a.h = h, a.wdt = wdt, a.relu = relu;
  a.K = cin * 9;
  a.ksteps = (a.K + 1) / 2;
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
  if (cin == 3 && cout <= 64) return launch_k<14, 2>(a, lds_bytes, s);
  if (cin == 1 && cout <= 64) return launch_k<5, 2>(a, lds_bytes, s);
  return launch_k<0, 2>(a, lds_bytes, s);
}

}  // namespace mv
