// sepstream.hip -- separable Gaussian for LARGE kernels (8 < K <= 63) as a row-streaming kernel (gfx950).
//
// gaussian_blur_image with a big kernel (the reference's ElasticTransform blurs its displacement field with
// K = int(8*sigma+1)|1, up to 41x41: transforms/v2/_geometry.py:1054-1075; _misc.py:147-155) runs here as the
// fused (1 x kx) then (ky x 1) pair of the primitive, like separable.hip / sepfast.hip, bit-identical to
// oracle/oracle.c's orc_separable_blur_f32.  The LDS-tile kernel (separable.hip) recomputes ky-1 halo rows of the row
// pass for every 16-row tile (2.4x redundant work at K = 23) and fits one workgroup per CU; this kernel never
// recomputes a row:
//   * a wave owns a 64*PX-pixel column segment and streams down a tall strip of rows;
//   * ROW PASS: the lane's PX pixels go to a wave-private LDS row buffer next to the K/2-pixel halos (fetched by a
//     few lanes, reflect-101 resolved per element); every lane then slides a window over it (16-byte LDS reads);
//   * COLUMN PASS: systolic fma chain in registers, `acc[k+1] = fma(w[k+1], t, acc[k])`: each new row-pass row is one
//     tap of the KB pending output rows, taps arrive in ascending order (the oracle's order), no row is ever stored
//     or recomputed.  KB is the register budget of the chain (15, 31 or 63 stages); a kernel with ky < KB taps is
//     front-padded with zero taps, which are exact no-ops.
#include <cstdlib>

#include "mv_common.h"

namespace mv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct StreamArgs {
  const void* x;  // float or uint8 storage (uint8: .to(float32) on load, round_() + narrow on store)
  void* y;
  Taps1D t;      // t.x: row taps zero-padded SYMMETRICALLY to KB; t.y: column taps FRONT-padded with zeros to KB
  int h, w, kx, ky;
  int rows, strips, col_segs;
  unsigned nblocks;
  long long nitems;
};

template <typename T>
__device__ inline void ss_load(const T* p, int px, float (&v)[4]) {
  if constexpr (sizeof(T) == 4) {
    if (px == 4) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(p);
      v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
    } else {
      const f32x2 q = *reinterpret_cast<const f32x2*>(p);
      v[0] = q.x, v[1] = q.y;
    }
  } else {
    if (px == 4) {
      const unsigned q = *reinterpret_cast<const unsigned*>(p);
      v[0] = (float)(q & 0xffu), v[1] = (float)((q >> 8) & 0xffu), v[2] = (float)((q >> 16) & 0xffu), v[3] = (float)(q >> 24);
    } else {
      const unsigned short q = *reinterpret_cast<const unsigned short*>(p);
      v[0] = (float)(q & 0xffu), v[1] = (float)(q >> 8);
    }
  }
}

template <typename T, int KB, int PX>
__global__ __launch_bounds__(256) void k_sepstream(const StreamArgs A) {
  constexpr int SEG = kWave * PX;          // pixels per wave segment
  constexpr int LMAX = 32;                 // halo capacity per side (K <= 63 -> R <= 31)
  constexpr int BUF = LMAX + SEG + LMAX + 8;  // floats per wave row buffer (+8: window over-read of the last lane)
  __shared__ __attribute__((aligned(16))) float rowbuf[4][BUF];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long item = (long long)xcd_remap(blockIdx.x, A.nblocks) * 4 + wave;
  if (item >= A.nitems) return;
  const int seg = (int)(item % A.col_segs);
  const long long t2 = item / A.col_segs;
  const int strip = (int)(t2 % A.strips);
  const long long plane = t2 / A.strips;
  const int h = A.h, w = A.w, kx = A.kx, ky = A.ky;
  const int rx = kx >> 1, ry = ky >> 1;
  const int Lr = (rx + 3) & ~3;            // halo rounded up to 4 (window chunks stay 16-byte aligned)
  const int xs0 = seg * SEG, xs = xs0 + lane * PX;
  const int y0 = strip * A.rows, y1 = min(y0 + A.rows, h);
  const size_t poff = (size_t)plane * h * w;
  const T* xp = static_cast<const T*>(A.x) + poff;
  T* yp = static_cast<T*>(A.y) + poff;
  float* rb = rowbuf[wave];
  // zero taps of the padded row kernel multiply whatever sits in the buffer beyond the real halo: keep it finite
  for (int i = lane; i < BUF; i += kWave) rb[i] = 0.f;

  float wy[KB];
#pragma unroll
  for (int i = 0; i < KB; ++i) wy[i] = A.t.y[i];

  float acc[KB - 1][PX];
#pragma unroll
  for (int k = 0; k < KB - 1; ++k)
#pragma unroll
    for (int p = 0; p < PX; ++p) acc[k][p] = 0.f;

  // halo duty: lanes [0, Lr/4) fetch the left halo, lanes [Lr/4, Lr/2) the right one (4 columns each)
  const int nh = Lr >> 2;
  const bool halo_l = lane < nh, halo_r = lane >= nh && lane < 2 * nh;
  const int hcol = halo_l ? xs0 - Lr + 4 * lane : xs0 + SEG + 4 * (lane - nh);  // first of my 4 halo columns
  float* hdst = halo_l ? rb + (LMAX - Lr) + 4 * lane : rb + LMAX + SEG + 4 * (lane - nh);

  const int t_first = y0 - ry, t_last = y1 - 1 + ry;
  // raw loads run one row ahead of the arithmetic
  f32x4 nv = {0.f, 0.f, 0.f, 0.f};
  float nhalo[4] = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int t) {
    const T* rowp = xp + (size_t)reflect_clamp(t, h) * w;
    nv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (xs < w) {
      float e[4] = {0.f, 0.f, 0.f, 0.f};
      ss_load<T>(rowp + xs, PX, e);
      nv = (f32x4){e[0], e[1], e[2], e[3]};
    }
    if (halo_l || halo_r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) nhalo[i] = (float)rowp[reflect_clamp(hcol + i, w)];
    }
  };
  fetch(t_first);
  for (int t = t_first; t <= t_last; ++t) {
    const T* rowp = xp + (size_t)reflect_clamp(t, h) * w;
    const f32x4 v = nv;
    const float hv[4] = {nhalo[0], nhalo[1], nhalo[2], nhalo[3]};
    if (t < t_last) fetch(t + 1);
    // ---- raw row -> wave-private LDS row buffer (own pixels + halos)
    if (PX == 4) {
      *reinterpret_cast<f32x4*>(rb + LMAX + lane * 4) = v;
    } else {
      *reinterpret_cast<f32x2*>(rb + LMAX + lane * 2) = (f32x2){v.x, v.y};
    }
    if (halo_l || halo_r) {
#pragma unroll
      for (int i = 0; i < 4; ++i) hdst[i] = hv[i];
    }
    // reflect-101 inside the segment's own span when the image ends inside it: columns >= w mirror to 2(w-1)-c
    if (xs0 + SEG > w) {
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        const int c = xs + p;
        if (c >= w && c < w + rx) rb[LMAX + lane * PX + p] = (float)rowp[reflect_clamp(c, w)];
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- row pass with the KB-tap zero-padded kernel: tmp[p] = sum_j wxp[j] * row[xs + p + j - RB]; the padding taps
    //      are exact no-ops in the fma chain, all indices are static (no window shuffling, taps in SGPRs)
    float tmp[PX];
    {
      constexpr int RB = KB / 2, RBA = (RB + 3) & ~3;
      constexpr int WIN = RBA + PX + RBA;              // floats fetched: columns xs - RBA .. xs + PX + RBA - 1
      float sg[WIN];
      const float* src = rb + LMAX + lane * PX - RBA;
      if (PX == 4) {
#pragma unroll
        for (int c = 0; c < WIN / 4; ++c) {
          const f32x4 q = *reinterpret_cast<const f32x4*>(src + 4 * c);
          sg[4 * c] = q.x, sg[4 * c + 1] = q.y, sg[4 * c + 2] = q.z, sg[4 * c + 3] = q.w;
        }
      } else {
#pragma unroll
        for (int c = 0; c < WIN / 2; ++c) {
          const f32x2 q = *reinterpret_cast<const f32x2*>(src + 2 * c);
          sg[2 * c] = q.x, sg[2 * c + 1] = q.y;
        }
      }
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        float a = fmaf(A.t.x[0], sg[RBA - RB + p], 0.f);
#pragma unroll
        for (int j = 1; j < KB; ++j) a = fmaf(A.t.x[j], sg[RBA - RB + p + j], a);
        tmp[p] = a;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- column pass (systolic): row t is tap KB-1 of output row t-ry, ..., tap 0 of the row KB-1 stages later
    float out[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
      out[p] = fmaf(wy[KB - 1], tmp[p], acc[KB - 2][p]);
#pragma unroll
      for (int k = KB - 2; k >= 1; --k) acc[k][p] = fmaf(wy[k], tmp[p], acc[k - 1][p]);
      acc[0][p] = fmaf(wy[0], tmp[p], 0.f);
    }
    const int oy = t - ry;
    if (t - t_first >= ky - 1 && xs < w) {  // the chain has seen all ky real taps of output row oy
      T* dst = yp + (size_t)oy * w + xs;
      if constexpr (sizeof(T) == 4) {
        if (PX == 4)
          __builtin_nontemporal_store((f32x4){out[0], out[1], out[2], out[3]}, reinterpret_cast<f32x4*>(dst));
        else
          __builtin_nontemporal_store((f32x2){out[0], out[1]}, reinterpret_cast<f32x2*>(dst));
      } else {
        unsigned pk = 0u;
#pragma unroll
        for (int p = 0; p < PX; ++p) pk = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf(out[p]), p, pk);  // round_() then narrow
        if (PX == 4)
          *reinterpret_cast<unsigned*>(dst) = pk;
        else
          *reinterpret_cast<unsigned short*>(dst) = (unsigned short)pk;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
bool sepstream_supported(const void* x, const void* y, bool u8, int h, int w, int kx, int ky) {
  const char* v = getenv("MV_FORCE_LDS_SEPARABLE");
  if (v && *v && *v != '0') return false;
  if (kx > 63 || ky > 63 || (kx <= 7 && ky <= 7)) return false;  // small kernels: sepfast / LDS tile
  if (h < 1 || w < 8) return false;
  const size_t es = u8 ? 1 : 4;
  if (kx <= 31 && ky <= 31) return (w % 4 == 0) && ((uintptr_t)x % (4 * es) == 0) && ((uintptr_t)y % (4 * es) == 0);
  return (w % 2 == 0) && ((uintptr_t)x % (2 * es) == 0) && ((uintptr_t)y % (2 * es) == 0);
}

template <typename T, int KB, int PX>
static int stream_launch(StreamArgs& a, int64_t planes, const float* k1d_x, const float* k1d_y, hipStream_t s) {
  for (int i = 0; i < KB; ++i) a.t.y[i] = 0.f, a.t.x[i] = 0.f;
  for (int i = 0; i < a.ky; ++i) a.t.y[KB - a.ky + i] = k1d_y[i];          // zero taps in front: exact no-ops
  for (int i = 0; i < a.kx; ++i) a.t.x[(KB - a.kx) / 2 + i] = k1d_x[i];    // centred: zero taps on both sides
  a.col_segs = (a.w + kWave * PX - 1) / (kWave * PX);
  int rows = 128;
  if (const char* e = getenv("MV_SEPSTREAM_ROWS")) rows = atoi(e) > 0 ? atoi(e) : rows;
  if (rows > a.h) rows = a.h;
  a.rows = rows;
  a.strips = (a.h + rows - 1) / rows;
  a.nitems = (long long)planes * a.strips * a.col_segs;
  if (a.nitems > 4LL * 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "separable: batch too large for one launch");
  a.nblocks = (unsigned)((a.nitems + 3) / 4);
  hipLaunchKernelGGL((k_sepstream<T, KB, PX>), dim3(a.nblocks), dim3(256), 0, s, a);
  return check_launch("k_sepstream");
}

int launch_sepstream(const void* x, void* y, bool u8, int64_t planes, int h, int w, const float* k1d_x, int kx,
                     const float* k1d_y, int ky, hipStream_t s) {
  StreamArgs a = {};
  a.x = x, a.y = y, a.h = h, a.w = w, a.kx = kx, a.ky = ky;
  const int kmax = kx > ky ? kx : ky;
  if (u8) {
    if (kmax <= 15) return stream_launch<uint8_t, 15, 4>(a, planes, k1d_x, k1d_y, s);
    if (kmax <= 31) return stream_launch<uint8_t, 31, 4>(a, planes, k1d_x, k1d_y, s);
    return stream_launch<uint8_t, 63, 2>(a, planes, k1d_x, k1d_y, s);
  }
  if (kmax <= 15) return stream_launch<float, 15, 4>(a, planes, k1d_x, k1d_y, s);
  if (kmax <= 31) return stream_launch<float, 31, 4>(a, planes, k1d_x, k1d_y, s);
  return stream_launch<float, 63, 2>(a, planes, k1d_x, k1d_y, s);
}

}  // namespace mv
