// linear_mfma.hip -- nn.Linear(k, m) [+bias] [+ReLU] of the small CNNs' classifier (models/vgg.py:42-50:
// 25088 -> 4096 -> 4096 -> num_classes) on the gfx950 fp32 matrix core.
//
// y[n][j] = relu(sum_k x[n][k] * W[j][k] + b[j]).  GEMM view: M = out features (A = W, row-major [M][K] as nn.Linear
// stores it), N = batch rows (B[k][n] = x[n][k]), fp32 v_mfma_f32_32x32x2_f32, ONE accumulator per output fed in
// ascending k with the bias as the last tap (A = bias, B = 1) -- bit-for-bit oracle/oracle.c's fmaf chain + bias.
// Single pass (no split-K): parallelism is (M/32) x (N/32) wave tiles, enough for training-size batches (>= 139 TFLOP/s at
// N = 256) -- but 25088 -> 4096 at batch 64 is 64 workgroups streaming 411 MB of weights at 0.36 TB/s.
// Sliced-K pass (launch_linear_sliced, caller-provided workspace): when the single-pass grid would leave most CUs idle,
// K is cut into S contiguous slices of `slice_len` (a multiple of 32); workgroup (m-block, n-block, slice) runs the same
// ascending-k chain over its slice from +0 and writes the raw partial sums to workspace[slice][n][m]; k_linear_reduce
// then adds the partials in ascending slice order (p0 + p1 + ... ), adds the bias and applies the ReLU.  The order is
// fixed by linear_plan() -- deterministic, no atomics -- and oracle/oracle.c restates it (orc_linear_sliced_*).
//
//   workgroup  128 features x up to 128 batch rows; wave w owns feature tile w (32 rows of W) and all batch tiles;
//   K loop     chunks of 32: W chunk -> LDS in MFMA fragment order (16-byte loads along k), x chunk -> LDS [n][33]
//              (odd pitch: the per-lane ds_read_b32 of column k is conflict-free); 16 k-steps per chunk.
#include <cstdlib>

#include "mv_common.h"
#include "mv_act.h"

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
  int rowfast;  // W staging items: row fastest over the lanes (LDS writes spread over 32 banks) or float4-of-a-row fastest (8-way
                // bank conflicts, but a wave reads 8 whole 128-byte lines per instruction instead of 16 bytes of 64 lines)
  int slices, chunks_per_slice;  // sliced-K pass: y is the workspace [slices][n][m], no bias / ReLU in the main kernel
};

// NT = batch tiles (of 32 rows) per workgroup.  The next chunk's global loads are issued into registers before the
// current chunk's MFMAs and written to LDS after them, so HBM/L2 latency hides behind the matrix pipe.
// (batch <= 4 runs on k_linear_gemv below.)
template <bool RELU, int NT, bool SLICED = false>
__global__ __launch_bounds__(256, 2) void k_linear(const LinArgs A) {
  __shared__ __attribute__((aligned(16))) float wfr[16 * 4 * 64];        // [s][m][lane]
  __shared__ __attribute__((aligned(16))) float xs[NT * 32 * kLPitch];    // [n][k], pitch 33
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hf = lane >> 5;
  const int K = A.k, M = A.m, N = A.n;
  const int mb = blockIdx.x % A.mblocks;
  const int rest = blockIdx.x / A.mblocks;
  const int nb = SLICED ? rest % A.nblocks_n : rest, slice = SLICED ? rest / A.nblocks_n : 0;
  const int ch_begin = SLICED ? slice * A.chunks_per_slice : 0;
  const int ch_end = SLICED ? min(ch_begin + A.chunks_per_slice, A.chunks) : A.chunks;
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
      const int row = A.rowfast ? (idx & 127) : (idx >> 3), q = A.rowfast ? (idx >> 7) : (idx & 7);  // W item -> (row, float4)
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
      const int row = A.rowfast ? (idx & 127) : (idx >> 3), q = A.rowfast ? (idx >> 7) : (idx & 7);  // W item -> (row, float4)
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

  gload(ch_begin);
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    __syncthreads();  // previous chunk fully consumed
    lstore();
    __syncthreads();
    if (ch + 1 < ch_end) gload(ch + 1);  // in flight while the MFMAs below run

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
  if (!SLICED && A.b != nullptr) {
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
        float* yr = A.y + ((size_t)slice * N + nn) * M;  // sliced: workspace[slice][n][m]
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f = j0 + 32 * wave + 8 * g + 4 * hf;
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = acc[t][4 * g + i];
            if (RELU) v[i] = relu_f32(v[i]);
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


// ---------------------------------------------------------------------------------------------- batch <= 4: weight streaming
// A 32 x 32 MFMA tile would spend 16 passes on 28-31 padding columns (25088 -> 4096 at batch 1: 42 us of matrix pipe for
// 0.2 GFLOP), so lane (hf, l31) of wave w runs feature 128 * block + 32 * w + l31 for batch rows hf and hf + 2 as plain v_fma_f32
// chains in the same ascending-k order (the oracle's fmaf chain, which is also what the MFMA computes): the job is to stream W.
// The first version staged ONE 16 KB chunk per workgroup at a time and read it back with a 4-byte LDS load per fma: 3.4 TB/s on
// the 411 MB of VGG's first classifier layer (Little: 4 workgroups x 16 KB in flight per CU).  Here
//   * stages of 64 k (32 KB of W per workgroup), TWO of them in flight in registers behind the one being consumed from LDS;
//   * W tile [feature row][k] at pitch 68 (16-byte reads of consecutive rows fall on distinct banks): one ds_read_b128 = 4 taps,
//     x rows read as 16-byte broadcasts;
//   * k past the slice's end is staged as zeros in BOTH operands (a +0 product at the end of the chain: an exact no-op).
// kGK = k per stage (64: 256-byte pieces of 128 weight rows per stage; 128: 512-byte pieces, half the workgroups per CU)
template <bool RELU, bool SLICED, int kGK = 64>
__global__ __launch_bounds__(256, 2) void k_linear_gemv(const LinArgs A) {
  constexpr int kGPitch = kGK + 4;  // W tile pitch (floats): pitch / 4 odd
  constexpr int QR = kGK / 4;       // float4 per row of a stage
  constexpr int WU = 128 * QR / 256;  // W float4 per thread and stage
  __shared__ __attribute__((aligned(16))) float wt[128 * kGPitch];
  __shared__ __attribute__((aligned(16))) float xt[4 * kGK];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hf = lane >> 5;
  const int K = A.k, M = A.m, N = A.n;
  const int mb = blockIdx.x % A.mblocks, slice = SLICED ? blockIdx.x / A.mblocks : 0;
  const int k_begin = SLICED ? slice * A.chunks_per_slice * kLK : 0;
  const int k_end = SLICED ? min(k_begin + A.chunks_per_slice * kLK, K) : K;
  const int j0 = mb * 128;
  const int stages = (k_end - k_begin + kGK - 1) / kGK;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // W item u of a thread: row (tid + 256 u) / 16, float4 (tid + 256 u) % 16 of the stage; x item (threads < 64): row tid / 16
  struct Regs { f32x4 w[WU]; f32x4 x; };
  auto gload = [&](int st, Regs& r) {  // addresses clamped into the tensors, values untouched until lstore (the loads stay in flight)
    const int kc = k_begin + st * kGK;
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int idx = tid + 256 * u, row = idx / QR, q = idx % QR;
      const bool ok = j0 + row < M && kc + 4 * q < k_end;
      r.w[u] = *reinterpret_cast<const f32x4*>(A.w + (ok ? (size_t)(j0 + row) * K + kc + 4 * q : 0));
    }
    const int row = (tid / QR) & 3, q = tid % QR;
    const bool ok = tid < 4 * QR && row < N && kc + 4 * q < k_end;
    r.x = *reinterpret_cast<const f32x4*>(A.x + (ok ? (size_t)row * K + kc + 4 * q : 0));
  };
  auto lstore = [&](int st, const Regs& r) {
    const int kc = k_begin + st * kGK;
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int idx = tid + 256 * u, row = idx / QR, q = idx % QR;
      const bool ok = j0 + row < M && kc + 4 * q < k_end;
      *reinterpret_cast<f32x4*>(wt + row * kGPitch + 4 * q) = ok ? r.w[u] : zero4;
    }
    if (tid < 4 * QR) {
      const int row = tid / QR, q = tid % QR;
      const bool ok = row < N && kc + 4 * q < k_end;
      *reinterpret_cast<f32x4*>(xt + row * kGK + 4 * q) = ok ? r.x : zero4;
    }
  };
  float gv[2] = {0.f, 0.f};
  auto consume = [&]() {
    const float* wp = wt + (32 * wave + l31) * kGPitch;
    const float* x0 = xt + hf * kGK;
    const float* x1 = xt + (hf + 2) * kGK;
#pragma unroll
    for (int q = 0; q < kGK / 4; ++q) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(wp + 4 * q);
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(x0 + 4 * q);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(x1 + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        gv[0] = fmaf(w4[i], a4[i], gv[0]);
        gv[1] = fmaf(w4[i], b4[i], gv[1]);
      }
    }
  };

  Regs r0, r1;
  if (stages > 0) gload(0, r0);
  if (stages > 1) gload(1, r1);
  for (int st = 0; st < stages; st += 2) {
    __syncthreads();  // the previous stage is consumed
    lstore(st, r0);
    __syncthreads();
    if (st + 2 < stages) gload(st + 2, r0);
    consume();
    if (st + 1 >= stages) break;
    __syncthreads();
    lstore(st + 1, r1);
    __syncthreads();
    if (st + 3 < stages) gload(st + 3, r1);
    consume();
  }

  const int j = j0 + 32 * wave + l31;
  if (j < M) {
    const float bias = (!SLICED && A.b != nullptr) ? A.b[j] : 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int nn = hf + 2 * r;
      if (nn < N) {
        float v = gv[r];
        if (!SLICED && A.b != nullptr) v = v + bias;
        if (RELU) v = relu_f32(v);
        A.y[((size_t)slice * N + nn) * M + j] = v;
      }
    }
  }
}

// W staging order (LinArgs::rowfast).  Measured (tools/perf_linear.py, profiles/r01_perf_linear_sliced_k_v2.log): row-fastest
// is 8-25 % faster wherever the MFMAs matter (25088 -> 4096 at batch 1024: 73 -> 87 TFLOP/s; batch 256: 752 -> 618 us)
// and on the smaller layers at any batch; only the purely weight-streaming case (k >= 16384 at batch <= 32) prefers whole
// 128-byte lines per instruction (126 against 132 us).
static bool linear_gemv() {
  const char* e = tune_env("MV_LINEAR_GEMV");  // tuning knob: 0 = MFMA tiles for every batch size
  return !(e && *e) || atoi(e) != 0;
}

static int linear_rowfast(int64_t n, int k) {
  const char* e = tune_env("MV_LINEAR_ROWFAST");  // tuning knob
  if (e && *e) return atoi(e) != 0;
  return !(n <= 32 && k >= 16384);
}

// y[n][j] = relu?(p[0][n][j] + p[1][n][j] + ... + p[S-1][n][j] + b[j]): partials added in ascending slice order
struct LinReduceArgs {
  const float* part;
  const float* b;
  float* y;
  long long total;  // n * m
  int m, slices, relu;
};

// The partial sums are LOADED eight slices at a time, then added in ascending order: one load-then-add per slice paid a memory
// round trip per slice (32 slices for VGG's first classifier layer at batch 1).
__global__ __launch_bounds__(256) void k_linear_reduce(const LinReduceArgs A) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= A.total) return;
  float acc = A.part[i];
  for (int s0 = 1; s0 < A.slices; s0 += 8) {
    float p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = A.part[(size_t)min(s0 + j, A.slices - 1) * A.total + i];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (s0 + j < A.slices) acc = acc + p[j];
  }
  if (A.b != nullptr) acc = acc + A.b[(int)(i % A.m)];
  if (A.relu) acc = relu_f32(acc);
  A.y[i] = acc;
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

// The K slicing of the workspace entry point: *slices >= 1 (1 = single pass, no workspace), *slice_len = k values per
// slice (a multiple of 32).  Sliced when the single-pass grid (128 features x 32 batch rows per workgroup) has fewer than
// 512 workgroups: enough slices for ~1024 workgroups, each at least 256 k long.
void linear_plan(int64_t n, int k, int m, int* slices, int* slice_len) {
  const long long chunks = (k + kLK - 1) / kLK;
  const long long base = (long long)((m + 127) / 128) * ((n + 31) / 32);
  long long want = 1;
  if (n > 0 && base < 512) {
    want = (1024 + base - 1) / base;
    if (want > chunks / 8) want = chunks / 8;
    if (want < 2) want = 1;
  }
  const long long cps = (chunks + want - 1) / want;
  *slices = (int)((chunks + cps - 1) / cps);
  *slice_len = (int)(cps * kLK);
}

int launch_linear_sliced(const float* x, const float* w, const float* b, float* y, int64_t n, int k, int m, int relu,
                         float* ws, hipStream_t s) {
  int slices, slice_len;
  linear_plan(n, k, m, &slices, &slice_len);
  if (slices <= 1) return launch_linear(x, w, b, y, n, k, m, relu, s);
  LinArgs a = {};
  a.x = x, a.w = w, a.b = nullptr, a.y = ws;
  a.n = (int)n, a.k = k, a.m = m, a.relu = 0;
  a.chunks = (k + kLK - 1) / kLK;
  a.mblocks = (m + 127) / 128;
  a.nblocks_n = (int)((n + 31) / 32);
  a.slices = slices, a.chunks_per_slice = slice_len / kLK;
  a.rowfast = linear_rowfast(n, k);
  a.vec_w = (k % 4 == 0) && ((uintptr_t)w % 16 == 0);
  a.vec_x = (k % 4 == 0) && ((uintptr_t)x % 16 == 0);
  a.vec_y = (m % 4 == 0) && ((uintptr_t)ws % 16 == 0);
  const long long nb = (long long)a.mblocks * a.nblocks_n * slices;  // < 512 * 128 by construction
  const bool gemv = n <= 4 && linear_gemv() && a.vec_w && a.vec_x;
  if (gemv && tune_env("MV_GEMV_K128"))
    hipLaunchKernelGGL((k_linear_gemv<false, true, 128>), dim3((unsigned)nb), dim3(256), 0, s, a);
  else if (gemv)
    hipLaunchKernelGGL((k_linear_gemv<false, true>), dim3((unsigned)nb), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_linear<false, 1, true>), dim3((unsigned)nb), dim3(256), 0, s, a);
  if (int rc = check_launch(gemv ? "k_linear_gemv (sliced)" : "k_linear (sliced)")) return rc;
  LinReduceArgs r = {};
  r.part = ws, r.b = b, r.y = y, r.total = (long long)n * m, r.m = m, r.slices = slices, r.relu = relu;
  hipLaunchKernelGGL(k_linear_reduce, dim3((unsigned)((r.total + 255) / 256)), dim3(256), 0, s, r);
  return check_launch("k_linear_reduce");
}

int launch_linear(const float* x, const float* w, const float* b, float* y, int64_t n, int k, int m, int relu,
                  hipStream_t s) {
  LinArgs a = {};
  a.x = x, a.w = w, a.b = b, a.y = y;
  a.n = (int)n, a.k = k, a.m = m, a.relu = relu;
  a.chunks = (k + kLK - 1) / kLK;
  a.mblocks = (m + 127) / 128;
  a.rowfast = linear_rowfast(n, k);
  a.vec_w = (k % 4 == 0) && ((uintptr_t)w % 16 == 0);
  a.vec_x = (k % 4 == 0) && ((uintptr_t)x % 16 == 0);
  a.vec_y = (m % 4 == 0) && ((uintptr_t)y % 16 == 0);
  if (n > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "linear: problem too large for one launch");
  // batch tiles per workgroup: fewer when the grid would otherwise leave CUs idle (no split-K: see the header)
  const long long tiles = (long long)a.mblocks * ((n + 31) / 32);
  if (n <= 4 && linear_gemv() && a.vec_w && a.vec_x) {
    a.nblocks_n = 1;
    if (a.relu)
      hipLaunchKernelGGL((k_linear_gemv<true, false>), dim3((unsigned)a.mblocks), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((k_linear_gemv<false, false>), dim3((unsigned)a.mblocks), dim3(256), 0, s, a);
    return check_launch("k_linear_gemv");
  }
  if (tiles <= 1024) return launch_linear_nt<1>(a, s);
  if (tiles <= 4096) return launch_linear_nt<2>(a, s);
  return launch_linear_nt<4>(a, s);
}

}  // namespace mv
