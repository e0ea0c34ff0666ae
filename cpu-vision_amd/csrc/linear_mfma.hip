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
constexpr int kLNT = 4;      // batch tiles (of 32) per workgroup

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

template <bool RELU>
__global__ __launch_bounds__(256, 2) void k_linear(const LinArgs A) {
  __shared__ __attribute__((aligned(16))) float wfr[16 * 4 * 64];          // [s][m][lane]
  __shared__ __attribute__((aligned(16))) float xs[kLNT * 32 * kLPitch];    // [n][k], pitch 65
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hf = lane >> 5;
  const int K = A.k, M = A.m, N = A.n;
  const int mb = blockIdx.x % A.mblocks, nb = blockIdx.x / A.mblocks;
  const int j0 = mb * 128, n0 = nb * (kLNT * 32);
  const int ntiles = min(kLNT, (N - n0 + 31) / 32);  // wave-uniform

  f32x16 acc[kLNT];
#pragma unroll
  for (int t = 0; t < kLNT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  for (int ch = 0; ch < A.chunks; ++ch) {
    const int kc = ch * kLK;
    __syncthreads();
    // ---- W chunk: 128 rows x 64 k -> fragment order
#pragma unroll 2
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;  // 1024 float4
      const int row = idx >> 3, q = idx & 7;
      const int j = j0 + row;
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (j < M) {
        const float* src = A.w + (size_t)j * K + kc + 4 * q;
        if (A.vec_w && kc + 4 * q + 3 < K) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src);
          e[0] = v.x, e[1] = v.y, e[2] = v.z, e[3] = v.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (kc + 4 * q + i < K) e[i] = src[i];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kl = 4 * q + i;
        wfr[(((kl >> 1) * 4 + (row >> 5)) << 6) + (kl & 1) * 32 + (row & 31)] = e[i];
      }
    }
    // ---- x chunk: up to 128 rows x 64 k
#pragma unroll 2
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + 256 * u;
      const int row = idx >> 3, q = idx & 7;
      const int nn = n0 + row;
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      if (row < ntiles * 32 && nn < N) {
        const float* src = A.x + (size_t)nn * K + kc + 4 * q;
        if (A.vec_x && kc + 4 * q + 3 < K) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src);
          e[0] = v.x, e[1] = v.y, e[2] = v.z, e[3] = v.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (kc + 4 * q + i < K) e[i] = src[i];
        }
      }
      if (row < ntiles * 32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xs[row * kLPitch + 4 * q + i] = e[i];
      }
    }
    __syncthreads();

    const float* ap = wfr + wave * 64 + lane;
    const float* bp = xs + l31 * kLPitch + hf;
#pragma unroll 8
    for (int s = 0; s < kLK / 2; ++s) {
      const float av = ap[s * 256];
#pragma unroll
      for (int t = 0; t < kLNT; ++t) {
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
    for (int t = 0; t < kLNT; ++t)
      if (t < ntiles) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
  }

  // ---- ReLU + store: lane <-> batch row, registers 4g..4g+3 <-> 4 consecutive features
#pragma unroll
  for (int t = 0; t < kLNT; ++t) {
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

int launch_linear(const float* x, const float* w, const float* b, float* y, int64_t n, int k, int m, int relu,
                  hipStream_t s) {
  LinArgs a = {};
  a.x = x, a.w = w, a.b = b, a.y = y;
  a.n = (int)n, a.k = k, a.m = m, a.relu = relu;
  a.chunks = (k + kLK - 1) / kLK;
  a.mblocks = (m + 127) / 128;
  a.nblocks_n = (int)((n + kLNT * 32 - 1) / (kLNT * 32));
  a.vec_w = (k % 4 == 0) && ((uintptr_t)w % 16 == 0);
  a.vec_x = (k % 4 == 0) && ((uintptr_t)x % 16 == 0);
  a.vec_y = (m % 4 == 0) && ((uintptr_t)y % 16 == 0);
  const long long nb = (long long)a.mblocks * a.nblocks_n;
  if (nb > 0x7fffffffLL || n > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "linear: problem too large for one launch");
  if (relu)
    hipLaunchKernelGGL(k_linear<true>, dim3((unsigned)nb), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(k_linear<false>, dim3((unsigned)nb), dim3(256), 0, s, a);
  return check_launch("k_linear");
}

}  // namespace mv
