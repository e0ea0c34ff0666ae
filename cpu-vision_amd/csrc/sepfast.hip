// sepfast.hip -- register-streaming separable Gaussian (K = 3, 5, 7) and the fused Gaussian -> Sobel graph
// (BASELINE cfg3) for gfx950.  Same contract and bit-exact results as the LDS kernel in separable.hip,
// which remains the general path (any K <= 63, any width / alignment).
//
// The reference would run cfg3 as three calls of its primitive pad(reflect) + conv2d(groups=C)
// (transforms/v2/functional/_misc.py:153-155): (1 x K), (K x 1), then the 3x3 Sobel pair -- four frame reads and
// four frame writes.  Here x is read once and gx, gy are written once (12 B / element), and nothing but
// registers sits in between:
//
//   * a wave owns a 256-pixel column segment (64 lanes x 4 pixels, 16-byte loads / stores) and walks down a
//     strip of rows, like k_dw3x3;
//   * ROW PASS in registers: the lane's 4 pixels plus HL = K/2 (+1 when Sobel follows) neighbours per side
//     (wave shuffles; lanes 0 / 63 fetch one aligned 16-byte halo per row).  With Sobel each lane evaluates
//     6 columns (its 4 and one to each side) so that the Sobel stage needs no second exchange;
//   * COLUMN PASS as a systolic chain: a new row-pass row t is tap dy of the K pending blurred rows
//     t+K/2 .. t-K/2; `acc[k+1] = fma(w[k+1], t, acc[k])` advances all of them with K fma per column and no
//     register moves, and feeds every blurred row its taps in ascending order (the oracle's order);
//   * SOBEL on a 3-row window of blurred rows; reflect-101 of the BLURRED image (blur(-1) := blur(1)) is resolved
//     by patching window rows / columns, never by blurring a reflected input, so the result is bit-identical to
//     composing the three primitive calls (oracle/oracle.c orc_gaussian_sobel_f32).
//
// Raw rows are fetched through the reflect map, so row-pass rows outside the image are recomputed from the
// mirrored row (bit-identical by construction).  Loads run G rows ahead of the arithmetic.
#include <cstdlib>

#include "mv_common.h"

namespace mv {

typedef float f4 __attribute__((ext_vector_type(4)));

#ifndef MV_SEPFAST_GROUP
#define MV_SEPFAST_GROUP 4
#endif
#ifndef MV_SEPFAST_ROWS_SOBEL
#define MV_SEPFAST_ROWS_SOBEL 48  // run time flat 16..48 (profiles/r03_sweep_sepfast_order.log: 32 -> 48 within 0.3-0.7 %), 64 slower; 48 re-reads
                                  // 6 halo rows per 48 instead of per 32 (FETCH_SIZE 1.06x -> 1.04x of the input)
#endif
#ifndef MV_SEPFAST_NT
#define MV_SEPFAST_NT 1  // non-temporal stores for gx / gy (0: plain stores -- tools/ab_alloc_lottery.py --variant)
#endif
#ifndef MV_SEPFAST_ROWS_BLUR
#define MV_SEPFAST_ROWS_BLUR 16   // 8/16: 6.0 TB/s, 32: 5.86, 64: 5.64: --op sep5
#endif

struct SepFastArgs {
  const float* x;
  float* y;   // blur only
  float* gx;  // sobel
  float* gy;
  Taps1D t;
  int h, w;
  int rows, strips, col_segs;
  int pgroup;       // planes per group of the (group, strip, plane in group, segment) item order
  long long planes;
  unsigned nblocks;
  long long nitems;
  FramePtrs fp;  // mv_*_v: per-frame base pointers for x / y (blur only; n == 0: contiguous batch)
};

struct RawF {
  f4 v;  // the lane's 4 pixels
  f4 e;  // lanes 0 / 63: the 4 pixels left of / right of the segment
};

__device__ inline RawF sf_load(const float* rowp, int xs, int w, int lane) {
  RawF q;
  q.v = (f4){0.f, 0.f, 0.f, 0.f};
  q.e = q.v;
  if (xs < w) q.v = *reinterpret_cast<const f4*>(rowp + xs);
  const int hx = (lane == 0) ? xs - 4 : xs + 4;
  const bool hl = (lane == 0 && xs > 0) || (lane == kWave - 1 && xs + 4 < w);
  if (hl) q.e = *reinterpret_cast<const f4*>(rowp + hx);
  return q;
}

// win[0 .. 4+2*HL): columns xs-HL .. xs+3+HL of one row, reflect-101 at the image's left / right edge.
template <int HL>
__device__ inline void sf_window(const RawF& q, int xs, int w, int lane, float (&win)[4 + 2 * HL]) {
  const float own[4] = {q.v.x, q.v.y, q.v.z, q.v.w};
  float lf[4], rt[4];  // lane-1's pixels (columns xs-4..xs-1), lane+1's pixels (columns xs+4..xs+7)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float e = (i == 0) ? q.e.x : (i == 1) ? q.e.y : (i == 2) ? q.e.z : q.e.w;
    // only the HL values next to the lane are ever used; the compiler drops the other shuffles
    const float up = __shfl_up(own[i], 1), dn = __shfl_down(own[i], 1);
    lf[i] = (lane == 0) ? e : up;
    rt[i] = (lane == kWave - 1) ? e : dn;
  }
  if (xs == 0) {  // columns -1, -2, -3, -4 -> 1, 2, 3, 4
    lf[3] = own[1], lf[2] = own[2], lf[1] = own[3], lf[0] = rt[0];
  }
  if (w - xs == 4) {  // columns w, w+1, w+2, w+3 -> w-2, w-3, w-4, w-5
    rt[0] = own[2], rt[1] = own[1], rt[2] = own[0], rt[3] = lf[3];
  }
#pragma unroll
  for (int i = 0; i < HL; ++i) win[i] = lf[4 - HL + i];
#pragma unroll
  for (int i = 0; i < 4; ++i) win[HL + i] = own[i];
#pragma unroll
  for (int i = 0; i < HL; ++i) win[HL + 4 + i] = rt[i];
}

template <int K, bool SOBEL>
__global__ __launch_bounds__(256) void k_sepfast(const SepFastArgs A) {
  constexpr int R = K / 2;
  constexpr int E = SOBEL ? 1 : 0;   // extra blurred columns per side
  constexpr int HL = R + E;          // raw neighbours per side
  constexpr int NC = 4 + 2 * E;      // blurred columns a lane evaluates
  constexpr int G = MV_SEPFAST_GROUP;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  const long long item = (long long)xcd_remap(blockIdx.x, A.nblocks) * 4 + wave;
  if (item >= A.nitems) return;
  // item order (group of `pgroup` planes, strip, plane in the group, segment).  pgroup = 1 (the product's choice) is plain (plane,
  // strip, segment) order: an XCD's share of the grid walks consecutive strips of one plane -- one contiguous 8 MB window of HBM.
  // pgroup = planes / 8 turns an XCD's concurrent workgroups into ONE strip level of its planes, so that the K - 1 + 2 halo rows
  // two adjacent strips share are re-read within a fraction of a strip's run time (L2 hits instead of the 6 % extra FETCH_SIZE of
  // profiles/r03_pmc_traffic_configs.txt) -- and measured 2.7 % SLOWER on cfg3 (profiles/r03_sweep_sepfast_order.log): twelve
  // planes' rows 33 MB apart cost more in DRAM locality than the halo re-reads do.  Taller strips lose the same way.
  const long long per_group = (long long)A.pgroup * A.strips * A.col_segs;
  const long long grp = item / per_group;
  const long long r1 = item - grp * per_group;
  const int pig = (int)min((long long)A.pgroup, A.planes - grp * A.pgroup);  // planes in this group (the last one may be short)
  const int level = pig * A.col_segs;
  const int strip = (int)(r1 / level);
  const int r2 = (int)(r1 - (long long)strip * level);
  const long long plane = grp * A.pgroup + r2 / A.col_segs;
  const int seg = r2 % A.col_segs;
  const int h = A.h, w = A.w;
  const int xs = seg * 256 + lane * 4;
  const int y0 = strip * A.rows, y1 = min(y0 + A.rows, h);  // output rows [y0, y1)
  const size_t poff = (size_t)plane * h * w;
  const float* xp = frame_in<float>(A.fp, A.x, plane, (size_t)h * w);
  float* const yplane = SOBEL ? nullptr : frame_out<float>(A.fp, A.y, plane, (size_t)h * w);

  float wx[K], wy[K];
#pragma unroll
  for (int i = 0; i < K; ++i) wx[i] = A.t.x[i], wy[i] = A.t.y[i];

  // blurred rows this strip needs, and the row-pass rows those need (reflect-mapped when outside the image)
  const int b0 = SOBEL ? max(y0 - 1, 0) : y0;
  const int b1 = SOBEL ? min(y1, h - 1) : y1 - 1;  // inclusive
  const int t0 = b0 - R, t1 = b1 + R;               // inclusive

  auto row_ptr = [&](int t) -> const float* { return xp + (size_t)reflect_clamp(t, h) * w; };

  float acc[K - 1 > 0 ? K - 1 : 1][NC];  // pending blurred rows (systolic column pass)
#pragma unroll
  for (int k = 0; k < K - 1; ++k)
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[k][c] = 0.f;
  float btop[NC], bmid[NC];  // blurred 3-row window (Sobel)
#pragma unroll
  for (int c = 0; c < NC; ++c) btop[c] = bmid[c] = 0.f;

  RawF nxt[G];
#pragma unroll
  for (int g = 0; g < G; ++g) nxt[g] = (t0 + g <= t1) ? sf_load(row_ptr(t0 + g), xs, w, lane) : RawF{};

  for (int tb = t0; tb <= t1; tb += G) {
    RawF cur[G];
#pragma unroll
    for (int g = 0; g < G; ++g) cur[g] = nxt[g];
    if (tb + G <= t1) {
#pragma unroll
      for (int g = 0; g < G; ++g)
        nxt[g] = (tb + G + g <= t1) ? sf_load(row_ptr(tb + G + g), xs, w, lane) : RawF{};
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int t = tb + g;  // wave-uniform; shuffles below run for every lane of the wave
      float win[4 + 2 * HL];
      sf_window<HL>(cur[g], xs, w, lane, win);
      if (t > t1) continue;
      // ---- row pass: tmp[c] = sum_dx wx[dx] * x[col(c) + dx - R],   col(c) = xs - E + c
      float tmp[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        float a = fmaf(wx[0], win[c], 0.f);
#pragma unroll
        for (int dx = 1; dx < K; ++dx) a = fmaf(wx[dx], win[c + dx], a);
        tmp[c] = a;
      }
      // ---- column pass (systolic): row t is tap K-1 of blurred row t-R, ..., tap 0 of blurred row t+R
      float blur[NC];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (K == 1) {
          blur[c] = fmaf(wy[0], tmp[c], 0.f);
        } else {
          blur[c] = fmaf(wy[K - 1], tmp[c], acc[K - 2][c]);
#pragma unroll
          for (int k = K - 2; k >= 1; --k) acc[k][c] = fmaf(wy[k], tmp[c], acc[k - 1][c]);
          acc[0][c] = fmaf(wy[0], tmp[c], 0.f);
        }
      }
      const int by = t - R;  // the blurred row completed by this step (valid once K rows have gone in)
      if (t - t0 < K - 1) continue;
      if constexpr (!SOBEL) {
        if (xs < w) {
          f4 v = {blur[0], blur[1], blur[2], blur[3]};
          __builtin_nontemporal_store(v, reinterpret_cast<f4*>(yplane + (size_t)by * w + xs));
        }
      } else {
        // reflect-101 of the blurred image at the left / right image edge
        if (xs == 0) blur[0] = blur[2];            // column -1 -> 1
        if (w - xs == 4) blur[5] = blur[3];        // column w  -> w-2
        // window (btop, bmid, blur) = blurred rows by-2, by-1, by  ->  output row by-1
        auto emit = [&](int oy, const float (&tp)[NC], const float (&md)[NC], const float (&bt)[NC]) {
          float ogx[4], ogy[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            // Sobel taps [[-1,0,1],[-2,0,2],[-1,0,1]] / transpose as the oracle's 9-tap fma chain from +0;
            // the zero taps are exact no-ops and the +-1 taps exact adds, so they are written as such
            float a = fmaf(-1.f, tp[p], 0.f);
            a = a + tp[p + 2];
            a = fmaf(-2.f, md[p], a);
            a = fmaf(2.f, md[p + 2], a);
            a = a - bt[p];
            a = a + bt[p + 2];
            ogx[p] = a;
            float b = fmaf(-1.f, tp[p], 0.f);
            b = fmaf(-2.f, tp[p + 1], b);
            b = b - tp[p + 2];
            b = b + bt[p];
            b = fmaf(2.f, bt[p + 1], b);
            b = b + bt[p + 2];
            ogy[p] = b;
          }
          if (xs < w) {
            const size_t o = poff + (size_t)oy * w + xs;
            f4 v1 = {ogx[0], ogx[1], ogx[2], ogx[3]}, v2 = {ogy[0], ogy[1], ogy[2], ogy[3]};
#if MV_SEPFAST_NT
            __builtin_nontemporal_store(v1, reinterpret_cast<f4*>(A.gx + o));
            __builtin_nontemporal_store(v2, reinterpret_cast<f4*>(A.gy + o));
#else
            *reinterpret_cast<f4*>(A.gx + o) = v1;
            *reinterpret_cast<f4*>(A.gy + o) = v2;
#endif
          }
        };
        const int oy = by - 1;
        if (oy >= y0 && oy < y1) {
          if (oy == 0)
            emit(oy, blur, bmid, blur);  // blurred row -1 := blurred row 1
          else
            emit(oy, btop, bmid, blur);
        }
        if (by == h - 1 && by >= y0 && by < y1) emit(by, bmid, blur, bmid);  // blurred row h := row h-2
#pragma unroll
        for (int c = 0; c < NC; ++c) btop[c] = bmid[c], bmid[c] = blur[c];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
static int sf_env_int(const char* name, int dflt) {
  const char* v = tune_env(name);
  return (v && *v) ? atoi(v) : dflt;
}

template <int K, bool SOBEL>
static int sf_launch(SepFastArgs& a, hipStream_t s) {
  hipLaunchKernelGGL((k_sepfast<K, SOBEL>), dim3(a.nblocks), dim3(256), 0, s, a);
  return check_launchf("k_sepfast<%d,%s>", K, SOBEL ? "sobel" : "blur");
}

bool sepfast_supported(const float* x, const float* o1, const float* o2, int h, int w, int kx, int ky, bool sobel) {
  if (kx != ky || (kx != 3 && kx != 5 && kx != 7)) return false;
  if (w % 4 != 0 || w < 8 || h < 8) return false;
  if ((uintptr_t)x % 16 || (uintptr_t)o1 % 16 || (o2 && (uintptr_t)o2 % 16)) return false;
  const char* v = tune_env("MV_FORCE_LDS_SEPARABLE");
  if (v && *v && *v != '0') return false;
  (void)sobel;
  return true;
}

int launch_sepfast(const float* x, float* y, float* gx, float* gy, bool sobel, int64_t planes, int h, int w,
                   const float* k1d_x, const float* k1d_y, int k, hipStream_t s) {
  SepFastArgs a = {};
  a.x = x, a.y = y, a.gx = gx, a.gy = gy, a.h = h, a.w = w;
  if (!sobel) fill_frames(a.fp);
  for (int i = 0; i < k; ++i) a.t.x[i] = k1d_x[i], a.t.y[i] = k1d_y[i];
  a.col_segs = (w + 255) / 256;
  int rows = sf_env_int("MV_SEPFAST_ROWS", sobel ? MV_SEPFAST_ROWS_SOBEL : MV_SEPFAST_ROWS_BLUR);
  if (rows < 1) rows = 1;
  if (rows > h) rows = h;
  a.rows = rows;
  a.strips = (h + rows - 1) / rows;
  a.nitems = (long long)planes * a.strips * a.col_segs;
  a.planes = planes;
  a.pgroup = sf_env_int("MV_SEPFAST_PGROUP", 1);  // tuning knob: see the kernel's comment on the item order
  if (a.pgroup < 1) a.pgroup = 1;
  if (a.nitems > 4LL * 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "separable: batch too large for one launch");
  a.nblocks = (unsigned)((a.nitems + 3) / 4);
  if (sobel) {
    if (k == 3) return sf_launch<3, true>(a, s);
    if (k == 5) return sf_launch<5, true>(a, s);
    return sf_launch<7, true>(a, s);
  }
  if (k == 3) return sf_launch<3, false>(a, s);
  if (k == 5) return sf_launch<5, false>(a, s);
  return sf_launch<7, false>(a, s);
}

}  // namespace mv
