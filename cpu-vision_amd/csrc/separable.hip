// separable.hip -- separable Gaussian and the fused Gaussian -> Sobel graph (BASELINE cfg3) on gfx950.
//
// The reference runs a Gaussian blur as ONE 2-D conv of the outer-product kernel
// (transforms/v2/functional/_misc.py:93-99,147-155).  `separable 5x5 then Sobel` (cfg3) is a new
// operator on the same primitive: three calls of pad(reflect)+conv2d(groups=C) -- (1 x kx), (ky x 1),
// then the 3x3 Sobel pair -- i.e. 4 frame reads + 4 frame writes if run unfused.  Here the whole
// graph is one kernel: x is read once from HBM, gx and gy are written once (36 B / pixel), every
// intermediate lives in LDS:
//
//   stage 1  raw tile + halo -> LDS      (16-byte loads, reflect-101 resolved while staging)
//   stage 2  row pass  (1 x kx)          raw  -> tmp   (LDS -> LDS, 16-byte ds reads, sliding window)
//   stage 3  column pass (ky x 1)        tmp  -> global (blur only)  or  -> blur tile (LDS)
//   stage 4  Sobel pair on the blur tile, reflect resolved by index mapping, -> global gx, gy
//
// Every stage uses the oracle's tap order (ascending, fma chain from +0), and the Sobel stage
// addresses the blurred tile through the reflect map (blur(-1) := blur(1)) instead of blurring a
// reflected input, so the result is bit-identical to composing the three primitive calls.
#include "mv_common.h"

namespace mv {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kTW = 256;

struct SepArgs {
  const float* x;
  float* y;   // blur output (blur only)
  float* gx;  // sobel outputs
  float* gy;
  Taps1D t;
  int h, w, ky, kx;
  int tiles_x, tiles_y;
  unsigned nblocks;
  FramePtrs fp;  // mv_*_v: per-frame base pointers for x / y (blur only; n == 0: contiguous batch)
};

__device__ inline float sob9(const float (&w)[9], float p0, float p1, float p2, float p3, float p4, float p5, float p6,
                             float p7, float p8) {
  float acc = fmaf(w[0], p0, 0.f);
  acc = fmaf(w[1], p1, acc);
  acc = fmaf(w[2], p2, acc);
  acc = fmaf(w[3], p3, acc);
  acc = fmaf(w[4], p4, acc);
  acc = fmaf(w[5], p5, acc);
  acc = fmaf(w[6], p6, acc);
  acc = fmaf(w[7], p7, acc);
  acc = fmaf(w[8], p8, acc);
  return acc;
}

template <bool SOBEL, int RPT, bool VEC>
__global__ __launch_bounds__(256) void k_separable(const SepArgs A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int ky = A.ky, kx = A.kx, ry = ky >> 1, rx = kx >> 1;
  const int h = A.h, w = A.w;
  constexpr int TH = 4 * RPT;
  constexpr int E = SOBEL ? 1 : 0;   // extra blurred rim needed by the Sobel stage
  constexpr int Lt = SOBEL ? 4 : 0;  // tmp / blur tiles: rim rounded up to 4 columns
  const int Lrx = (rx + 3) & ~3;     // row-pass window reach, rounded up to 4
  const int Lr = Lt + Lrx;           // raw tile halo (columns)
  const int pr = kTW + 2 * Lr;       // raw pitch
  constexpr int pt = kTW + 2 * Lt;   // tmp / blur pitch
  const int rows_r = TH + 2 * E + 2 * ry;
  constexpr int rows_b = TH + 2 * E;
  float* raw = lds;  // reused as the blur tile after stage 2 (rows_b*pt <= rows_r*pr)
  float* tmp = lds + rows_r * pr;

  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
  const unsigned wid = xcd_remap(blockIdx.x, A.nblocks);
  const int tx = wid % A.tiles_x;
  const unsigned t2 = wid / A.tiles_x;
  const int ty = t2 % A.tiles_y;
  const long long plane = t2 / A.tiles_y;
  const int x0 = tx * kTW, y0 = ty * TH;
  const float* xp = frame_in<float>(A.fp, A.x, plane, (size_t)h * w);

  // ---- stage 1: raw tile, rows [y0-E-ry, ...), cols [x0-Lr, ...), reflect in both directions
  {
    const int slots = pr >> 2;
    for (int idx = tid; idx < rows_r * slots; idx += 256) {
      const int row = idx / slots, slot = idx - row * slots;
      const int gx0 = x0 - Lr + (slot << 2);
      const float* rp = xp + (size_t)reflect_clamp(y0 - E - ry + row, h) * w;
      f4 v;
      if (VEC && gx0 >= 0 && gx0 + 3 < w) {
        v = *reinterpret_cast<const f4*>(rp + gx0);
      } else {
        v.x = rp[reflect_clamp(gx0 + 0, w)];
        v.y = rp[reflect_clamp(gx0 + 1, w)];
        v.z = rp[reflect_clamp(gx0 + 2, w)];
        v.w = rp[reflect_clamp(gx0 + 3, w)];
      }
      *reinterpret_cast<f4*>(raw + row * pr + (slot << 2)) = v;
    }
  }
  __syncthreads();

  // ---- stage 2: row pass.  tmp[row][c] (c = tile column incl. rim Lt) = sum_dx kx[dx] * raw[row][c + Lrx - rx + dx]
  {
    constexpr int slots = pt >> 2;
    const int skip = Lrx - rx;                    // leading window elements that no tap uses
    const int nch = 1 + 2 * (Lrx >> 2);           // 16-byte chunks per window
    for (int idx = tid; idx < rows_r * slots; idx += 256) {
      const int row = idx / slots, slot = idx - row * slots;
      const float* src = raw + row * pr + (slot << 2);  // window start = output column - Lrx (in raw coords: +Lt+Lrx-Lrx... )
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int c = 0; c < nch; ++c) {
        const f4 q = *reinterpret_cast<const f4*>(src + (c << 2));
        const float e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s0 = s1, s1 = s2, s2 = s3, s3 = e[k];
          const int dx = (c << 2) + k - skip - 3;
          if (dx >= 0 && dx < kx) {
            const float wv = A.t.x[dx];
            a0 = fmaf(wv, s0, a0), a1 = fmaf(wv, s1, a1), a2 = fmaf(wv, s2, a2), a3 = fmaf(wv, s3, a3);
          }
        }
      }
      f4 o = {a0, a1, a2, a3};
      *reinterpret_cast<f4*>(tmp + row * pt + (slot << 2)) = o;
    }
  }
  __syncthreads();

  if constexpr (!SOBEL) {
    // ---- stage 3 (blur only): column pass straight to global.  lane -> 4 columns, wave -> RPT rows
    float* yp = frame_out<float>(A.fp, A.y, plane, (size_t)h * w);
    const int ox = x0 + (lane << 2);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int lr = wave * RPT + r, oy = y0 + lr;
      if (oy < h) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int dy = 0; dy < ky; ++dy) {
          const f4 q = *reinterpret_cast<const f4*>(tmp + (lr + dy) * pt + (lane << 2));
          const float wv = A.t.y[dy];
          a0 = fmaf(wv, q.x, a0), a1 = fmaf(wv, q.y, a1), a2 = fmaf(wv, q.z, a2), a3 = fmaf(wv, q.w, a3);
        }
        float* rp = yp + (size_t)oy * w;
        if (VEC) {
          if (ox < w) {
            f4 v = {a0, a1, a2, a3};
            __builtin_nontemporal_store(v, reinterpret_cast<f4*>(rp + ox));
          }
        } else {
          if (ox + 0 < w) rp[ox + 0] = a0;
          if (ox + 1 < w) rp[ox + 1] = a1;
          if (ox + 2 < w) rp[ox + 2] = a2;
          if (ox + 3 < w) rp[ox + 3] = a3;
        }
      }
    }
  } else {
    // ---- stage 3 (fused): column pass into the blur tile (rows y0-1 .. y0+TH, cols x0-4 .. x0+TW+3)
    float* blur = raw;
    {
      constexpr int slots = pt >> 2;
      for (int idx = tid; idx < rows_b * slots; idx += 256) {
        const int row = idx / slots, slot = idx - row * slots;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int dy = 0; dy < ky; ++dy) {
          const f4 q = *reinterpret_cast<const f4*>(tmp + (row + dy) * pt + (slot << 2));
          const float wv = A.t.y[dy];
          a0 = fmaf(wv, q.x, a0), a1 = fmaf(wv, q.y, a1), a2 = fmaf(wv, q.z, a2), a3 = fmaf(wv, q.w, a3);
        }
        f4 o = {a0, a1, a2, a3};
        *reinterpret_cast<f4*>(blur + row * pt + (slot << 2)) = o;
      }
    }
    __syncthreads();

    // ---- stage 4: Sobel pair on the blurred image, reflect-101 by index mapping
    const float GX[9] = {-1.f, 0.f, 1.f, -2.f, 0.f, 2.f, -1.f, 0.f, 1.f};
    const float GY[9] = {-1.f, -2.f, -1.f, 0.f, 0.f, 0.f, 1.f, 2.f, 1.f};
    float* gxp = A.gx + (size_t)plane * h * w;
    float* gyp = A.gy + (size_t)plane * h * w;
    const int ox = x0 + (lane << 2);
    const int rem = w - ox;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int oy = y0 + wave * RPT + r;
      if (oy < h) {
        float win[3][6];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int gy = reflect_clamp(oy - 1 + k, h);
          const float* bp = blur + (gy - (y0 - 1)) * pt + (lane << 2);  // tile column of (ox - 4)
          const f4 ql = *reinterpret_cast<const f4*>(bp);
          const f4 qc = *reinterpret_cast<const f4*>(bp + 4);
          const f4 qr = *reinterpret_cast<const f4*>(bp + 8);
          float l = ql.w, a = qc.x, b = qc.y, c = qc.z, d = qc.w, rr = qr.x;
          if (ox == 0) l = b;   // column -1 -> 1
          if (rem == 4) rr = c;  // column w -> w-2
          if (rem == 3) d = b;
          if (rem == 2) c = a;
          if (rem == 1) b = l;
          win[k][0] = l, win[k][1] = a, win[k][2] = b, win[k][3] = c, win[k][4] = d, win[k][5] = rr;
        }
        float ogx[4], ogy[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          ogx[p] = sob9(GX, win[0][p], win[0][p + 1], win[0][p + 2], win[1][p], win[1][p + 1], win[1][p + 2], win[2][p],
                        win[2][p + 1], win[2][p + 2]);
          ogy[p] = sob9(GY, win[0][p], win[0][p + 1], win[0][p + 2], win[1][p], win[1][p + 1], win[1][p + 2], win[2][p],
                        win[2][p + 1], win[2][p + 2]);
        }
        float* rgx = gxp + (size_t)oy * w;
        float* rgy = gyp + (size_t)oy * w;
        if (VEC) {
          if (ox < w) {
            f4 v1 = {ogx[0], ogx[1], ogx[2], ogx[3]};
            f4 v2 = {ogy[0], ogy[1], ogy[2], ogy[3]};
            __builtin_nontemporal_store(v1, reinterpret_cast<f4*>(rgx + ox));
            __builtin_nontemporal_store(v2, reinterpret_cast<f4*>(rgy + ox));
          }
        } else {
#pragma unroll
          for (int p = 0; p < 4; ++p)
            if (ox + p < w) rgx[ox + p] = ogx[p], rgy[ox + p] = ogy[p];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <bool SOBEL, int RPT>
static size_t lds_bytes_for(int ky, int kx) {
  const int E = SOBEL ? 1 : 0, Lt = SOBEL ? 4 : 0;
  const int Lrx = ((kx / 2) + 3) & ~3;
  const int pr = kTW + 2 * (Lt + Lrx), pt = kTW + 2 * Lt;
  const int rows_r = 4 * RPT + 2 * E + 2 * (ky / 2);
  return (size_t)rows_r * (pr + pt) * sizeof(float);
}

template <bool SOBEL, int RPT>
static int launch_cfg(SepArgs& a, int64_t planes, bool vec, hipStream_t s) {
  const size_t lds_bytes = lds_bytes_for<SOBEL, RPT>(a.ky, a.kx);
  a.tiles_x = (a.w + kTW - 1) / kTW;
  a.tiles_y = (a.h + 4 * RPT - 1) / (4 * RPT);
  const long long nb = (long long)planes * a.tiles_x * a.tiles_y;
  if (nb > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "separable: batch too large for one launch");
  a.nblocks = (unsigned)nb;
  dim3 grid(a.nblocks), block(256);
  if (vec) {
    auto k = k_separable<SOBEL, RPT, true>;
    if (lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
    hipLaunchKernelGGL(k, grid, block, lds_bytes, s, a);
  } else {
    auto k = k_separable<SOBEL, RPT, false>;
    if (lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes);
    hipLaunchKernelGGL(k, grid, block, lds_bytes, s, a);
  }
  return check_launch("k_separable");
}

int launch_separable(const float* x, float* y, float* gx, float* gy, bool sobel, int64_t planes, int h, int w,
                     const float* k1d_x, int kx, const float* k1d_y, int ky, hipStream_t s) {
  SepArgs a = {};
  a.x = x, a.y = y, a.gx = gx, a.gy = gy;
  if (!sobel) fill_frames(a.fp);
  a.h = h, a.w = w, a.ky = ky, a.kx = kx;
  for (int i = 0; i < kx; ++i) a.t.x[i] = k1d_x[i];
  for (int i = 0; i < ky; ++i) a.t.y[i] = k1d_y[i];
  bool vec = (w % 4 == 0) && ((uintptr_t)x % 16 == 0);
  if (sobel)
    vec = vec && ((uintptr_t)gx % 16 == 0) && ((uintptr_t)gy % 16 == 0);
  else
    vec = vec && ((uintptr_t)y % 16 == 0);
  // taller tiles amortise the (ky-1)-row halo; fall back to 16-row tiles when 32 rows would not leave
  // room for two workgroups per CU (160 KiB LDS)
  if (sobel) {
    if (lds_bytes_for<true, 8>(ky, kx) <= 80 * 1024) return launch_cfg<true, 8>(a, planes, vec, s);
    if (lds_bytes_for<true, 4>(ky, kx) <= 160 * 1024) return launch_cfg<true, 4>(a, planes, vec, s);
  } else {
    if (lds_bytes_for<false, 8>(ky, kx) <= 80 * 1024) return launch_cfg<false, 8>(a, planes, vec, s);
    if (lds_bytes_for<false, 4>(ky, kx) <= 160 * 1024) return launch_cfg<false, 4>(a, planes, vec, s);
  }
  return set_error(MV_ERR_UNSUPPORTED, "separable: %dx%d taps exceed the LDS tile", ky, kx);
}

}  // namespace mv
