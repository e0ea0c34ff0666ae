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
//     or recomputed.  KB is the register budget of the chain (15, 23, 31, 47 or 63 stages); a kernel with ky < KB taps
//     is front-padded with zero taps, which are exact no-ops.
//   * TAPS: 2*KB wave-uniform floats do not fit the SGPR file next to the addressing state (the compiler spilled them
//     to VGPR lanes: one v_readlane per fma).  They stay in the kernel-argument segment instead and are streamed
//     through a two-deep SGPR ring, 8*S taps at a time, by s_load_dwordx8 issued one step ahead of its use (scalar
//     cache hits, no VALU or LDS cost).
#include <cstddef>
#include <cstdlib>
#include <utility>

#include "mv_common.h"

namespace mv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef float f32x8 __attribute__((ext_vector_type(8)));
// any width, any alignment: gfx950 global memory takes these at any byte address (tools/micro/unaligned.hip); the lane at a
// ragged right edge (w % PX != 0) loads / stores its 1..PX-1 pixels one by one
typedef float f32x4a __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2a __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned int u32a __attribute__((aligned(1)));
typedef unsigned short u16a __attribute__((aligned(1)));
typedef const char __attribute__((address_space(4))) * kernarg_ptr;

struct StreamArgs {
  const void* x;  // float or uint8 storage (uint8: .to(float32) on load, round_() + narrow on store)
  void* y;
  // taps[0..KBP): row taps zero-padded SYMMETRICALLY to KB; taps[KBP..2*KBP): column taps FRONT-padded with zeros to
  // KB and stored REVERSED (the systolic chain consumes them from tap KB-1 down); KBP = KB + 1, a multiple of 8
  float taps[128];
  int h, w, kx, ky;
  int rows, strips, col_segs;
  unsigned nblocks;
  long long nitems;
  FramePtrs fp;  // mv_*_v: per-frame base pointers (n == 0: contiguous batch)
  TieList* ties;  // uint8 storage: lane-rows within tie_thresh of a rounding tie are appended (null: no check)
  float tie_thresh;
};

// the lane's PX pixels starting at p; `avail` = pixels left in the row from p (>= 1)
template <typename T>
__device__ inline void ss_load(const T* p, int px, int avail, float (&v)[4]) {
  if (avail < px) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < avail) v[i] = (float)p[i];
    return;
  }
  if constexpr (sizeof(T) == 4) {
    if (px == 4) {
      const f32x4a q = *reinterpret_cast<const f32x4a*>(p);
      v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
    } else {
      const f32x2a q = *reinterpret_cast<const f32x2a*>(p);
      v[0] = q.x, v[1] = q.y;
    }
  } else {
    if (px == 4) {
      const unsigned q = *reinterpret_cast<const u32a*>(p);
      v[0] = (float)(q & 0xffu), v[1] = (float)((q >> 8) & 0xffu), v[2] = (float)((q >> 16) & 0xffu), v[3] = (float)(q >> 24);
    } else {
      const unsigned short q = *reinterpret_cast<const u16a*>(p);
      v[0] = (float)(q & 0xffu), v[1] = (float)(q >> 8);
    }
  }
}

// uint8 storage: the loaded bytes stay RAW in the prefetch ring and become floats when the row is consumed.  (ss_load converts
// right behind the load -- a use: the compiler waited for the data there and the ring prefetched nothing; uint8 23 x 23 ran 54 %
// slower than fp32 on the same arithmetic.)  Bytes of a lane at the ragged right edge are gathered one by one (those lanes only).
__device__ inline unsigned ss_load_raw_u8(const uint8_t* p, int px, int avail) {
  if (avail < px) {
    unsigned q = 0u;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if (i < avail) q |= (unsigned)p[i] << (8 * i);
    return q;
  }
  if (px == 4) return *reinterpret_cast<const u32a*>(p);
  return *reinterpret_cast<const u16a*>(p);
}

// one group of 8 taps: kernel-argument segment -> SGPRs.  volatile: stays inside the row loop, in program order.
// (base + byte offset in an SGPR: a constant after unrolling, rematerialised by one s_mov instead of a live pointer)
__device__ inline f32x8 tap_load8(kernarg_ptr p, int byte_off) {
  f32x8 r;
  asm volatile("s_load_dwordx8 %0, %1, %2" : "=&s"(r) : "s"(p), "s"(byte_off));
  return r;
}
// every s_load issued so far has landed (the compiler's own lgkmcnt bookkeeping does not see them; extra outstanding
// scalar loads only make its LDS waits more conservative)
__device__ inline void tap_wait(f32x8& r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r)); }
__device__ inline void tap_pin(f32x8& r) { asm volatile("" : "+s"(r)); }

template <int PX>
__device__ inline void chain_pin(float (&a)[PX]) {
  if constexpr (PX == 4) {
    f32x4 v = {a[0], a[1], a[2], a[3]};
    asm volatile("" : "+v"(v));
    a[0] = v.x, a[1] = v.y, a[2] = v.z, a[3] = v.w;
  } else {
    f32x2 v = {a[0], a[1]};
    asm volatile("" : "+v"(v));
    a[0] = v.x, a[1] = v.y;
  }
}

template <typename F, int... I>
__device__ inline void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ inline void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// S = groups of 8 taps per ring step; PF = rows of raw loads in flight per wave
template <typename T, int KB, int PX, int S, int PF>
__global__ __launch_bounds__(256) void k_sepstream(const StreamArgs A) {
  constexpr int KBP = KB + 1;
  static_assert(KBP % (8 * S) == 0, "padded tap count must be a whole number of ring steps");
  constexpr int NS = KBP / (8 * S);        // ring steps per pass
  const kernarg_ptr ktaps = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(StreamArgs, taps);
  constexpr int SEG = kWave * PX;          // pixels per wave segment
  constexpr int LMAX = 32;                 // halo capacity per side (K <= 63 -> R <= 31)
  constexpr int BUF = LMAX + SEG + LMAX + 8;  // floats per wave row buffer (+8: window over-read of the last lane)
  __shared__ __attribute__((aligned(16))) float rowbuf[4][BUF];
  // uint8 storage: flagged lane-rows of each wave and strip (128 rows x 64 lanes x the 3.7 % (K = 23) .. 12 % (K = 63) the bound flags)
  constexpr int kTieCap = KB >= 47 ? 2048 : 1024;
  __shared__ unsigned long long tie_lds[sizeof(T) == 1 ? 4 * kTieCap : 1];
  TieWave tw = {tie_lds + (sizeof(T) == 1 ? (threadIdx.x >> 6) * kTieCap : 0), 0, kTieCap - 64, 0};  // the last 64 entries: tie_push's dump slots
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
  const T* xp = frame_in<T>(A.fp, A.x, plane, (size_t)h * w);
  T* yp = frame_out<T>(A.fp, A.y, plane, (size_t)h * w);
  float* rb = rowbuf[wave];
  // zero taps of the padded row kernel multiply whatever sits in the buffer beyond the real halo: keep it finite
  for (int i = lane; i < BUF; i += kWave) rb[i] = 0.f;

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
  // raw loads run PF rows ahead of the arithmetic, in a register ring: with 2 waves per SIMD (the chain's registers)
  // one row per wave in flight is ~2 MB chip-wide, far below HBM's bandwidth x latency product
  constexpr bool U8 = sizeof(T) == 1;
  f32x4 nv[PF];              // fp32 storage: the lane's pixels
  unsigned nraw[PF];         // uint8 storage: the lane's bytes, raw
  float nhalo[PF][4];        // fp32 storage
  unsigned nhraw[PF][4];     // uint8 storage: one byte each, raw
  auto fetch = [&](int t, auto slot) {
    constexpr int sl = decltype(slot)::value;
    const T* rowp = xp + (size_t)reflect_clamp(t, h) * w;
    if constexpr (U8) {
      nraw[sl] = 0u;
      if (xs < w) nraw[sl] = ss_load_raw_u8(reinterpret_cast<const uint8_t*>(rowp) + xs, PX, w - xs);
      if (halo_l || halo_r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) nhraw[sl][i] = reinterpret_cast<const uint8_t*>(rowp)[reflect_clamp(hcol + i, w)];
      }
    } else {
      nv[sl] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (xs < w) {
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        ss_load<T>(rowp + xs, PX, w - xs, e);
        nv[sl] = (f32x4){e[0], e[1], e[2], e[3]};
      }
      if (halo_l || halo_r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) nhalo[sl][i] = (float)rowp[reflect_clamp(hcol + i, w)];
      }
    }
  };
  static_for<PF>([&](auto r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) nhalo[decltype(r)::value][i] = 0.f, nhraw[decltype(r)::value][i] = 0u;
    nv[decltype(r)::value] = (f32x4){0.f, 0.f, 0.f, 0.f};
    nraw[decltype(r)::value] = 0u;
    if (t_first + decltype(r)::value <= t_last) fetch(t_first + decltype(r)::value, r);
  });
  f32x8 cur[S];
#pragma unroll
  for (int i = 0; i < S; ++i) cur[i] = tap_load8(ktaps, 32 * i);
  auto row_step = [&](const int t, auto slot) {
    constexpr int sl = decltype(slot)::value;
    const T* rowp = xp + (size_t)reflect_clamp(t, h) * w;
    f32x4 v = nv[sl];
    float hv[4] = {nhalo[sl][0], nhalo[sl][1], nhalo[sl][2], nhalo[sl][3]};
    if constexpr (U8) {  // decode here, a ring's length behind the loads
      const unsigned q = nraw[sl];
      v = (f32x4){(float)(q & 0xffu), (float)((q >> 8) & 0xffu), (float)((q >> 16) & 0xffu), (float)(q >> 24)};
#pragma unroll
      for (int i = 0; i < 4; ++i) hv[i] = (float)nhraw[sl][i];
    }
    if (t + PF <= t_last) fetch(t + PF, slot);
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
      for (int p = 0; p < PX; ++p) tmp[p] = 0.f;
#pragma unroll
      for (int q = 0; q < NS; ++q) {
        tap_wait(cur[0]);
#pragma unroll
        for (int i = 1; i < S; ++i) tap_pin(cur[i]);
        f32x8 nxt[S];
#pragma unroll
        for (int i = 0; i < S; ++i) nxt[i] = tap_load8(ktaps, 32 * ((q + 1) * S + i));
#pragma unroll
        for (int i = 0; i < S; ++i) tap_pin(cur[i]);  // the fmas below read `cur` after the issue above
#pragma unroll
        for (int i = 0; i < S; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int j = 8 * (q * S + i) + e;
            if (j < KB) {
#pragma unroll
              for (int p = 0; p < PX; ++p) tmp[p] = fmaf(cur[i][e], sg[RBA - RB + p + j], tmp[p]);
            }
          }
#pragma unroll
        for (int i = 0; i < S; ++i) cur[i] = nxt[i];
        __builtin_amdgcn_sched_barrier(0);  // this step's fmas stay between the issue of the next group and its wait
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- column pass (systolic): row t is tap KB-1 of output row t-ry, ..., tap 0 of the row KB-1 stages later;
    //      updated in place from the top of the chain down, so every stage still reads its predecessor's old value
    float out[PX];
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      tap_wait(cur[0]);
#pragma unroll
      for (int i = 1; i < S; ++i) tap_pin(cur[i]);
      f32x8 nxt[S];
#pragma unroll
      for (int i = 0; i < S; ++i) nxt[i] = tap_load8(ktaps, 32 * (((NS + q + 1) % (2 * NS)) * S + i));
#pragma unroll
      for (int i = 0; i < S; ++i) tap_pin(cur[i]);
#pragma unroll
      for (int i = 0; i < S; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int k = KB - 1 - (8 * (q * S + i) + e);
          const float wk = cur[i][e];
          if (k == KB - 1) {
#pragma unroll
            for (int p = 0; p < PX; ++p) out[p] = fmaf(wk, tmp[p], acc[KB - 2][p]);
          } else if (k >= 1) {
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[k][p] = fmaf(wk, tmp[p], acc[k - 1][p]);
          } else if (k == 0) {
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[0][p] = fmaf(wk, tmp[p], 0.f);
          }
          // pin the stage to this step: without it the compiler sinks the whole chain update below the last wait
          // (only `out` is used before the loop latch) and the ring degenerates into back-to-back load + wait
          if (k >= 0 && k < KB - 1) chain_pin<PX>(acc[k]);
        }
#pragma unroll
      for (int i = 0; i < S; ++i) cur[i] = nxt[i];
      __builtin_amdgcn_sched_barrier(0);
    }
    const int oy = t - ry;
    bool row_tie = false;  // uint8 storage: this lane's pixels of the row hold a value within tie_thresh of a rounding tie
    if (t - t_first >= ky - 1 && xs < w) {  // the chain has seen all ky real taps of output row oy
      T* dst = yp + (size_t)oy * w + xs;
      const bool full = xs + PX <= w;
      if constexpr (sizeof(T) == 4) {
        if (!full) {
#pragma unroll
          for (int p = 0; p < PX - 1; ++p)
            if (xs + p < w) dst[p] = out[p];
        } else if (PX == 4) {
          __builtin_nontemporal_store((f32x4a){out[0], out[1], out[2], out[3]}, reinterpret_cast<f32x4a*>(dst));
        } else {
          __builtin_nontemporal_store((f32x2a){out[0], out[1]}, reinterpret_cast<f32x2a*>(dst));
        }
      } else {
        unsigned pk = 0u;
        float tie_far = 0.f;  // the largest |v - rint(v)| of the lane's pixels (0.5 = exactly on a rounding tie)
#pragma unroll
        for (int p = 0; p < PX; ++p) {
          const float r = __builtin_rintf(out[p]);  // round_() then narrow
          pk = __builtin_amdgcn_cvt_pk_u8_f32(r, p, pk);
          if (xs + p < w) tie_far = fmaxf(tie_far, fabsf(out[p] - r));
        }
        row_tie = tie_far > A.tie_thresh;
        if (!full) {
#pragma unroll
          for (int p = 0; p < PX - 1; ++p)
            if (xs + p < w) dst[p] = (T)(pk >> (8 * p));
        } else if (PX == 4) {
          *reinterpret_cast<u32a*>(dst) = pk;
        } else {
          *reinterpret_cast<u16a*>(dst) = (unsigned short)pk;
        }
      }
    }
    if constexpr (sizeof(T) == 1) {  // no branch here (tie_push, mv_common.h): without a list nothing is flagged and nothing flushed
      tie_push(tw, row_tie && A.ties != nullptr, ((unsigned long long)plane * h + (unsigned)max(oy, 0)) * w + xs, lane);
    }
  };
  for (int t = t_first; t <= t_last; t += PF) {
    static_for<PF>([&](auto r) {
      if (t + decltype(r)::value <= t_last) row_step(t + decltype(r)::value, r);
    });
  }
  if constexpr (sizeof(T) == 1) {
    if (A.ties != nullptr) tie_flush(A.ties, tw, lane, blockIdx.x * 4u + (unsigned)wave);
  }
}

// ---------------------------------------------------------------------------------------------
bool sepstream_supported(const void* x, const void* y, bool u8, int h, int w, int kx, int ky) {
  const char* v = tune_env("MV_FORCE_LDS_SEPARABLE");
  if (v && *v && *v != '0') return false;
  if (kx > 63 || ky > 63 || (kx <= 7 && ky <= 7)) return false;  // small kernels: sepfast / LDS tile
  if (h < 1 || w < 8) return false;
  (void)x, (void)y, (void)u8;
  return true;
}

template <typename T, int KB, int PX, int S>
static int stream_launch_pf(StreamArgs& a, int64_t planes, const float* k1d_x, const float* k1d_y, hipStream_t s) {
  constexpr int KBP = KB + 1;
  for (int i = 0; i < 128; ++i) a.taps[i] = 0.f;
  for (int i = 0; i < a.kx; ++i) a.taps[(KB - a.kx) / 2 + i] = k1d_x[i];   // centred: zero taps on both sides
  // column tap k of the front-padded chain is k1d_y[k - (KB - ky)] (zero taps in front: exact no-ops); stored reversed
  for (int i = 0; i < a.ky; ++i) a.taps[KBP + (KB - 1) - (KB - a.ky + i)] = k1d_y[i];
  a.col_segs = (a.w + kWave * PX - 1) / (kWave * PX);
  // strip height: each strip re-reads K-1 halo rows, so tall strips -- but one halving (64 rows) when the launch would
  // otherwise have fewer than ~4k waves (256 x 3 x 224 x 224, K = 23: 210 -> 147 us; 32 rows is slower again)
  int rows = 128;
  if (planes * ((a.h + rows - 1) / rows) * a.col_segs < 4096) rows = 64;
  if (const char* e = tune_env("MV_SEPSTREAM_ROWS")) rows = atoi(e) > 0 ? atoi(e) : rows;
  if (rows > a.h) rows = a.h;
  a.rows = rows;
  a.strips = (a.h + rows - 1) / rows;
  a.nitems = (long long)planes * a.strips * a.col_segs;
  if (a.nitems > 4LL * 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "separable: batch too large for one launch");
  a.nblocks = (unsigned)((a.nitems + 3) / 4);
  int pf = KB <= 23 ? 2 : 4;  // measured (profiles/r01_perf_separable.log): deeper rings cost a wave per SIMD below KB = 31
  if (const char* e = tune_env("MV_SEPSTREAM_PF")) pf = atoi(e);
  if (pf <= 1)
    hipLaunchKernelGGL((k_sepstream<T, KB, PX, S, 1>), dim3(a.nblocks), dim3(256), 0, s, a);
  else if (pf <= 2)
    hipLaunchKernelGGL((k_sepstream<T, KB, PX, S, 2>), dim3(a.nblocks), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((k_sepstream<T, KB, PX, S, 4>), dim3(a.nblocks), dim3(256), 0, s, a);
  return check_launch("k_sepstream");
}

int sepstream_u8_pixels_per_lane(int kx, int ky) { return (kx > ky ? kx : ky) <= 31 ? 4 : 2; }

int launch_sepstream(const void* x, void* y, bool u8, int64_t planes, int h, int w, const float* k1d_x, int kx,
                     const float* k1d_y, int ky, hipStream_t s, TieList* ties, float tie_thresh) {
  StreamArgs a = {};
  a.x = x, a.y = y, a.h = h, a.w = w, a.kx = kx, a.ky = ky;
  a.ties = u8 ? ties : nullptr, a.tie_thresh = tie_thresh;
  fill_frames(a.fp);
  const int kmax = kx > ky ? kx : ky;
  if (u8) {
    if (kmax <= 15) return stream_launch_pf<uint8_t, 15, 4, 1>(a, planes, k1d_x, k1d_y, s);
    if (kmax <= 23) return stream_launch_pf<uint8_t, 23, 4, 1>(a, planes, k1d_x, k1d_y, s);
    if (kmax <= 31) return stream_launch_pf<uint8_t, 31, 4, 2>(a, planes, k1d_x, k1d_y, s);
    if (kmax <= 47) return stream_launch_pf<uint8_t, 47, 2, 1>(a, planes, k1d_x, k1d_y, s);
    return stream_launch_pf<uint8_t, 63, 2, 1>(a, planes, k1d_x, k1d_y, s);
  }
  if (kmax <= 15) return stream_launch_pf<float, 15, 4, 1>(a, planes, k1d_x, k1d_y, s);
  if (kmax <= 23) return stream_launch_pf<float, 23, 4, 1>(a, planes, k1d_x, k1d_y, s);
  if (kmax <= 31) return stream_launch_pf<float, 31, 4, 2>(a, planes, k1d_x, k1d_y, s);
  if (kmax <= 47) return stream_launch_pf<float, 47, 2, 1>(a, planes, k1d_x, k1d_y, s);
  return stream_launch_pf<float, 63, 2, 1>(a, planes, k1d_x, k1d_y, s);
}

}  // namespace mv
