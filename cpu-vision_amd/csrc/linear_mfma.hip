// linear_mfma.hip -- nn.Linear(k, m) [+bias] [+ReLU] of the small CNNs' classifier (models/vgg.py:42-50:
// 25088 -> 4096 -> 4096 -> num_classes) on the gfx950 fp32 matrix core.
//
// y[n][j] = relu(sum_k x[n][k] * W[j][k] + b[j]).  GEMM view: M = out features (A = W, row-major [M][K] as nn.Linear
// stores it), N = batch rows (B[k][n] = x[n][k]), fp32 v_mfma_f32_32x32x2_f32, ONE accumulator per output fed in
// ascending k with the bias as the last tap (A = bias, B = 1) -- bit-for-bit oracle/oracle.c's fmaf chain + bias.
// Deliberately no split-K (it would change the rounding order): parallelism is (M/32) x (N/32) wave tiles, enough
// for training-size batches (>= 139 TFLOP/s at N = 256), latency-bound for batch 1.
//
//   workgroup  128 features x up to 128 batch rows; wave w owns feature tile w (32 rows of W) and all batch tiles;
//   K loop     chunks of 32: W chunk -> LDS in MFMA fragment order (16-byte loads along k), x chunk -> LDS [n][33]
//              (odd pitch: the per-lane ds_read_b32 of column k is conflict-free); 16 k-steps per chunk.
#include <cstdlib>

#include "mv_common.h"

namespace mv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLK = 32;      // k per chunk (16 k-steps)
constexpr int kLPitch = 33;  // x tile pitch (odd: conflict-free column reads)

struct LinArgs {
  const float* x;
  const float* w;
  const float* b;
  float* y;
  int n, k, m;
  int chunks;
  int mblocks, nblocks_n;
  int relu, vec_w, vec_x, vec_y;
};

// NT = batch tiles (of 32 rows) per workgroup.  The next chunk's global loads are issued into registers before the
// current chunk's MFMAs and written to LDS after them, so HBM/L2 latency hides behind the matrix pipe.
template <bool RELU, int NT>
__global__ __launch_bounds__(256, 2) void k_linear(const LinArgs A) {
  __shared__ __attribute__((aligned(16))) float wfr[16 * 4 * 64];        // [s][m][lane]
  __shared__ __attribute__((aligned(16))) float xs[NT * 32 * kLPitch];    // [n][k], pitch 33
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hf = lane >> 5;
  const int K = A.k, M = A.m, N = A.n;
  const int mb = blockIdx.x % A.mblocks, nb = blockIdx.x / A.mblocks;
  const int j0 = mb * 128, n0 = nb * (NT * 32);
  const int ntiles = min(NT, (N - n0 + 31) / 32);  // wave-uniform
  constexpr int XU = NT;  // float4 per thread for the x chunk: NT*32 rows x 8 float4 / 256 threads

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  f32x4 wreg[4], xreg[XU];
  auto gload = [&](int ch) {
    const int kc = ch * kLK;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;  // 1024 float4 = 128 rows x 8
      const int row = idx >> 3, q = idx & 7;
      const int j = j0 + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (j < M) {
        const float* src = A.w + (size_t)j * K + kc + 4 * q;
        if (A.vec_w && kc + 4 * q + 3 < K) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
          if (kc + 4 * q + 0 < K) v.x = src[0];
          if (kc + 4 * q + 1 < K) v.y = src[1];
          if (kc + 4 * q + 2 < K) v.z = src[2];
          if (kc + 4 * q + 3 < K) v.w = src[3];
        }
      }
      wreg[u] = v;
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx >> 3, q = idx & 7;
      const int nn = n0 + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (nn < N) {
        const float* src = A.x + (size_t)nn * K + kc + 4 * q;
        if (A.vec_x && kc + 4 * q + 3 < K) {
          v = *reinterpret_cast<const f32x4*>(src);
        } else {
          if (kc + 4 * q + 0 < K) v.x = src[0];
          if (kc + 4 * q + 1 < K) v.y = src[1];
          if (kc + 4 * q + 2 < K) v.z = src[2];
          if (kc + 4 * q + 3 < K) v.w = src[3];
        }
      }
      xreg[u] = v;
    }
  };
  auto lstore = [&]() {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx >> 3, q = idx & 7;
      const float e[4] = {wreg[u].x, wreg[u].y, wreg[u].z, wreg[u].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kl = 4 * q + i;
        wfr[(((kl >> 1) * 4 + (row >> 5)) << 6) + (kl & 1) * 32 + (row & 31)] = e[i];
      }
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx >> 3, q = idx & 7;
      const float e[4] = {xreg[u].x, xreg[u].y, xreg[u].z, xreg[u].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) xs[row * kLPitch + 4 * q + i] = e[i];
    }
  };

  gload(0);
  for (int ch = 0; ch < A.chunks; ++ch) {
    __syncthreads();  // previous chunk fully consumed
    lstore();
    __syncthreads();
    if (ch + 1 < A.chunks) gload(ch + 1);  // in flight while the MFMAs below run

    const float* ap = wfr + wave * 64 + lane;
    const float* bp = xs + l31 * kLPitch + hf;
#pragma unroll 8
    for (int s = 0; s < kLK / 2; ++s) {
      const float av = ap[s * 256];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t < ntiles) {
          const float bv = bp[t * 32 * kLPitch + 2 * s];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
        }
      }
    }
  }

  // ---- bias as the last tap
  if (A.b != nullptr) {
    const int j = j0 + 32 * wave + l31;
    const float av = (hf == 0 && j < M) ? A.b[j] : 0.f;
    const float bv = hf ? 0.f : 1.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < ntiles) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
  }

  // ---- ReLU + store: lane <-> batch row, registers 4g..4g+3 <-> 4 consecutive features
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < ntiles) {
      const int nn = n0 + 32 * t + l31;
      if (nn < N) {
        float* yr = A.y + (size_t)nn * M;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f = j0 + 32 * wave + 8 * g + 4 * hf;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = acc[t][4 * g + i];
            if (RELU) v[i] = (v[i] < 0.f) ? 0.f : v[i];
          }
          if (A.vec_y && f + 3 < M) {
            *reinterpret_cast<f32x4*>(yr + f) = (f32x4){v[0], v[1], v[2], v[3]};
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (f + i < M) yr[f + i] = v[i];
          }
        }
      }
    }
  }
}

template <int NT>
static int launch_linear_nt(LinArgs& a, hipStream_t s) {
  a.nblocks_n = (a.n + NT * 32 - 1) / (NT * 32);
  const long long nb = (long long)a.mblocks * a.nblocks_n;
  if (nb > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "linear: problem too large for one launch");
  if (a.relu)
    hipLaunchKernelGGL((k_linear<true, NT>), dim3((unsigned)nb), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_linear<false, NT>), dim3((unsigned)nb), dim3(256), 0, s, a);
  return check_launch("k_linear");
}

int launch_linear(const float* x, const float* w, const float* b, float* y, int64_t n, int k, int m, int relu,
                  hipStream_t s) {
  LinArgs a = {};
  a.x = x, a.w = w, a.b = b, a.y = y;
  a.n = (int)n, a.k = k, a.m = m, a.relu = relu;
  a.chunks = (k + kLK - 1) / kLK;
  a.mblocks = (m + 127) / 128;
  a.vec_w = (k % 4 == 0) && ((uintptr_t)w % 16 == 0);
  a.vec_x = (k % 4 == 0) && ((uintptr_t)x % 16 == 0);
  a.vec_y = (m % 4 == 0) && ((uintptr_t)y % 16 == 0);
  if (n > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "linear: problem too large for one launch");
  // batch tiles per workgroup: fewer when the grid would otherwise leave CUs idle (no split-K: see the header)
  const long long tiles = (long long)a.mblocks * ((n + 31) / 32);
  if (tiles <= 1024) return launch_linear_nt<1>(a, s);
  if (tiles <= 4096) return launch_linear_nt<2>(a, s);
  return launch_linear_nt<4>(a, s);
}

}  // namespace mv
