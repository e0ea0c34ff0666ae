// conv3x3_c3.hip -- the first CNN layer proper: nn.Conv2d(3, cout<=64, 3, padding=1) [+bias] [+ReLU]
// (models/vgg.py:81-85 with in_channels = 3; BASELINE cfg4), tuned for gfx950's store path.
//
// Same arithmetic as conv3x3_mfma.hip (implicit GEMM on v_mfma_f32_32x32x2_f32, K = 27 taps in (ci,dy,dx) order
// + the bias as tap 27 with B = 1, so every output is the oracle's fmaf chain followed by `+ bias`, bit for bit).
// What differs is the shape of the work, chosen from measurements of the previous kernel (profiles/
// r01_tune_conv_ablation_nostore.log): with one pixel per lane every accumulator register left the CU as a
// 4-byte-per-lane store, 32 store instructions per 32x64 output tile, and those store instructions -- not HBM,
// not the MFMA pipe -- set the run time (0.66 ms with them, 0.47 ms without, L2-resident destination no faster).
// Here each lane owns FOUR consecutive pixels of a 128-pixel group (pixel 4*l + a belongs to MFMA tile a), so the
// same data leaves as 16-byte-per-lane stores: 4x fewer store instructions, 512 contiguous bytes per half-wave.
//
//   * a workgroup stages a band of `th` rows (all cin planes, 1-pixel zero halo) in LDS; the band's pixels are
//     numbered row-major ("flattened": W <= 256 -> whole rows, groups may straddle rows, no tail waste at W = 224)
//     or per 128/256-column tile for wider images;
//   * waves split M: waves {0,1} produce channels 0-31, waves {2,3} channels 32-63, each walking every other
//     128-pixel group; a group is 4 MFMA tiles x 14 k-steps = 56 MFMAs into 4 independent accumulators;
//   * B operands come from 18 ds_read_b128 per group (9 (ci,dy) rows x 8 floats: pixels 4l-1 .. 4l+6 cover all
//     3 taps of all 4 tiles); the per-half k selection (lanes 0-31: k = 2s, lanes 32-63: k = 2s+1) is a register
//     select, so no per-tap address arithmetic remains;
//   * two accumulator sets: while one group's chain runs, the previous group's 64 values per lane are ReLU'd and
//     leave as 16 dwordx4 stores, spread between the MFMAs (one basic block when the band is full).
#include <cstdlib>

#include "mv_common.h"
#include "mv_act.h"

namespace mv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef MV_C3_NT
#define MV_C3_NT 0  // plain stores: the write-only stream of this layer runs 5-10 % faster than with non-temporal stores (profiles/r02_tune_conv_c3_plain_stores.log, tools/micro/store_pattern.hip)
#endif
#ifndef MV_C3_ABLATE_MFMA
#define MV_C3_ABLATE_MFMA 0  // profiling builds only (wrong results): 1 = no MFMA, 2 = no stores
#endif

struct C3Args {
  const float* x;
  const float* w;
  const float* b;
  float* y;
  int cout, h, wdt;
  int mtiles;       // 1 or 2
  int th;           // rows per band
  int wc;           // band width in pixels (== wdt in flattened mode, else a multiple of 128)
  int pitch;        // LDS row pitch in floats (multiple of 4, >= wc + 2 + 8)
  int tiles_x, tiles_y;
  int groups;       // 128-pixel groups per band = ceil(th * wc / 128)
  int relu;
  int vec_rows;
  unsigned nblocks;
  // IN_U8: the input is uint8 and the preset's ToDtype(float, scale=True) + Normalize(mean, std) is applied while
  // staging (transforms/_presets.py:58-60): v = (u8 * (1/255) - mean[c]) / std[c]; the zero halo is added after it.
  float mean[3], stdv[3];
};

constexpr int kKS = 14;  // k-steps: K = 27 taps + bias slot
constexpr int kK = 27;

__device__ constexpr int c3_row(int k) { return k / 3; }  // (ci, dy) row index 0..8
__device__ constexpr int c3_dx(int k) { return k % 3; }

template <bool RELU>
__device__ inline float c3_act(float v) {
  if (RELU) v = relu_f32(v);  // NaN passes through, like torch.relu
  return v;
}

// FULL: every lane of every group maps to a pixel inside the image (no store predicate).
template <bool RELU, bool FULL, bool FULLM, bool IN_U8>
__global__ __launch_bounds__(256, 2) void k_conv3x3_c3(const C3Args A) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hf = lane >> 5;
  const int cout = A.cout, h = A.h, w = A.wdt;
  const int th = A.th, wc = A.wc, pitch = A.pitch;
  const int tile_rows = th + 2;
  float* xin = lds;                               // [3][th+2][pitch]
  float* wfr = lds + 3 * tile_rows * pitch + 16;  // [mtiles][14][64] A fragments (+16: over-read slack of xin)

  const unsigned wid = xcd_remap(blockIdx.x, A.nblocks);
  const int tx = wid % A.tiles_x;
  const unsigned t2 = wid / A.tiles_x;
  const int ty = t2 % A.tiles_y;
  const long long img = t2 / A.tiles_y;
  const int xb = tx * wc, yb = ty * th;
  const float* xp = A.x + (size_t)img * 3 * h * w;
  const uint8_t* xpu = reinterpret_cast<const uint8_t*>(A.x) + (size_t)img * 3 * h * w;  // IN_U8 view
  const size_t plane = (size_t)h * w;
  auto norm = [&](unsigned byte, int ci) -> float {
    const float v = (float)byte * (float)(1.0 / 255.0);
    return (v - A.mean[ci]) / A.stdv[ci];
  };

  // ---- A fragments: wfr[(m*14 + s)*64 + l] = W[32m + (l&31)][k = 2s + (l>>5)], k == 27 -> bias
  for (int q = wave; q < A.mtiles * kKS; q += 4) {
    const int s = q % kKS, m = q / kKS;
    const int co = 32 * m + l31, k = 2 * s + hf;
    float v = 0.f;
    if (co < cout && k < kK) v = A.w[(size_t)co * kK + k];
    if (co < cout && k == kK && A.b != nullptr) v = A.b[co];
    wfr[q * 64 + lane] = v;
  }
  // ---- zero-padded band: rows yb-1 .. yb+th, columns xb-1 .. xb+wc (tile column c <-> gx = xb - 1 + c)
  {
    constexpr int kStage = 8;
    const int cols = wc + 2;
    const int rows_total = 3 * tile_rows;
    if (A.vec_rows) {
      const int nq = (wc + 8) >> 2;
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
            if (q < nq && gy >= 0 && gy < h && gx0 >= 0 && gx0 + 3 < w) {
              if constexpr (IN_U8) {
                const unsigned b4 = *reinterpret_cast<const unsigned*>(xpu + ((size_t)ci * h + gy) * w + gx0);
                v[u] = (f32x4){norm(b4 & 0xffu, ci), norm((b4 >> 8) & 0xffu, ci), norm((b4 >> 16) & 0xffu, ci), norm(b4 >> 24, ci)};
              } else {
                v[u] = *reinterpret_cast<const f32x4*>(xp + ((size_t)ci * h + gy) * w + gx0);
              }
            }
          }
        }
#pragma unroll
        for (int u = 0; u < kStage; ++u) {
          const int it = base + 4 * u;
          if (it < nitems) {
            const int r = it / per_row, q = (it - r * per_row) * kWave + lane;
            if (q < nq) {
              float* dst = xin + r * pitch + (4 * q - 3);
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
            if (c < cols && gy >= 0 && gy < h && gx >= 0 && gx < w) {
              if constexpr (IN_U8)
                v[u] = norm(xpu[((size_t)ci * h + gy) * w + gx], ci);
              else
                v[u] = xp[((size_t)ci * h + gy) * w + gx];
            }
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

  // ---- roles: which 32-channel tile, which groups
  const int m = (A.mtiles == 2) ? (wave >> 1) : 0;
  const int gfirst = (A.mtiles == 2) ? (wave & 1) : wave;
  const int gstep = (A.mtiles == 2) ? 2 : 4;
  const int ng = (A.groups > gfirst) ? (A.groups - gfirst + gstep - 1) / gstep : 0;
  if (ng == 0) return;

  float afr[kKS];
#pragma unroll
  for (int s = 0; s < kKS; ++s) afr[s] = wfr[(m * kKS + s) * 64 + lane];

  // row offsets of the nine (ci, dy) rows inside the band (uniform)
  int rowoff[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) rowoff[r] = ((r / 3) * tile_rows + (r % 3)) * pitch;

  char* const simg = reinterpret_cast<char*>(A.y + (size_t)img * cout * plane);
  const unsigned chan_off = (unsigned)((size_t)(32 * m + 4 * hf) * plane * sizeof(float));

  // per-group lane geometry: flattened pixel f = 128*g + 4*l31 -> (ly, lx); LDS base; output byte offset
  auto geom = [&](int j, int& lbase, unsigned& voff, bool& ok) {
    const int g = gfirst + gstep * j;
    const unsigned f = 128u * (unsigned)g + 4u * (unsigned)l31;
    const unsigned ly = f / (unsigned)wc, lx = f - ly * (unsigned)wc;
    lbase = (int)(ly * (unsigned)pitch + lx);  // tile column lx <-> pixel (xb + lx) - 1: floats 0..7 = pixels -1..+6
    const int gy = yb + (int)ly, gx = xb + (int)lx;
    ok = FULL || (gy < h && gx < w && (int)ly < th);
    voff = chan_off + (unsigned)(((size_t)gy * w + gx) * sizeof(float));
  };
  auto load_row = [&](f32x4 (&dst)[2], int lbase, int r) {
    const float* p = xin + lbase + rowoff[r];
    dst[0] = *reinterpret_cast<const f32x4*>(p);
    dst[1] = *reinterpret_cast<const f32x4*>(p + 4);
  };
  auto elem = [](const f32x4 (&q)[2], int i) -> float {  // i in 0..7, compile-time after unrolling
    const f32x4& v = q[i >> 2];
    return (i & 3) == 0 ? v.x : (i & 3) == 1 ? v.y : (i & 3) == 2 ? v.z : v.w;
  };
  // drain accumulator register i of the previous group: 4 consecutive pixels of channel 32m + cu(i) + 4hf
  auto drain = [&](const f32x16 (&acc)[4], int i, unsigned voff, bool ok) {
    const int cu = (i & 3) + 8 * (i >> 2);
    f32x4 v = {c3_act<RELU>(acc[0][i]), c3_act<RELU>(acc[1][i]), c3_act<RELU>(acc[2][i]), c3_act<RELU>(acc[3][i])};
    f32x4* dst = reinterpret_cast<f32x4*>(simg + (size_t)cu * plane * sizeof(float) + voff);
    if (MV_C3_ABLATE_MFMA == 2) {
      if (v.x == 12345.678f) *dst = v;
    } else if ((FULL || ok) && (FULLM || 32 * m + cu + 4 * hf < cout)) {
#if MV_C3_NT
      __builtin_nontemporal_store(v, dst);
#else
      *dst = v;
#endif
    }
  };
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  f32x16 accA[4], accB[4];
  f32x4 row[9][2];    // the nine (ci,dy) rows of the current group (live range: a couple of k-steps each)
  f32x4 nrow[2][2];   // rows 0, 1 of the next group, fetched near the end of the current one
  int lb_cur, lb_next;
  unsigned vo_cur, vo_prev = 0;
  bool ok_cur, ok_prev = false;
  geom(0, lb_cur, vo_cur, ok_cur);
  load_row(nrow[0], lb_cur, 0);
  load_row(nrow[1], lb_cur, 1);

  // One group: 14 k-steps x 4 tiles into `cur`; rows are fetched one row ahead of their first use; the previous
  // group (`prev`) is drained one accumulator register every third MFMA slot; rows 0,1 of the next group are
  // fetched at steps 10 and 12.
#define MV_C3_GROUP(cur, prev, j, have_prev)                                                             \
  {                                                                                                      \
    unsigned vo_n = 0;                                                                                   \
    bool ok_n = false;                                                                                   \
    geom(min((j) + 1, ng - 1), lb_next, vo_n, ok_n);                                                      \
    row[0][0] = nrow[0][0], row[0][1] = nrow[0][1], row[1][0] = nrow[1][0], row[1][1] = nrow[1][1];       \
    _Pragma("unroll") for (int s = 0; s < kKS; ++s) {                                                    \
      /* rows needed by this step: c3_row(2s) and c3_row(2s+1); keep one row of lookahead */              \
      constexpr_for_rows(s)                                                                              \
      _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                                    \
        const int k0 = 2 * s, k1 = 2 * s + 1;                                                            \
        const float b0 = elem(row[c3_row(k0)], a + c3_dx(k0));                                           \
        const float b1 = (k1 < kK) ? elem(row[c3_row(k1 < kK ? k1 : 0)], a + c3_dx(k1)) : 1.0f;           \
        const float bsel = hf ? b1 : b0;                                                                 \
        if (MV_C3_ABLATE_MFMA == 1) { if (s == 0) cur[a] = zero; cur[a][s] += afr[s] * bsel; }           \
        else cur[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[s], bsel, s == 0 ? zero : cur[a], 0, 0, 0); \
        if (have_prev) {                                                                                 \
          const int slot = 4 * s + a;                                                                    \
          if (slot % 3 == 2 && slot / 3 < 16) drain(prev, slot / 3, vo_prev, ok_prev);                    \
        }                                                                                                \
      }                                                                                                  \
      if (s == 10) load_row(nrow[0], lb_next, 0);                                                        \
      if (s == 12) load_row(nrow[1], lb_next, 1);                                                        \
    }                                                                                                    \
    vo_prev = vo_cur, ok_prev = ok_cur;                                                                  \
    lb_cur = lb_next, vo_cur = vo_n, ok_cur = ok_n;                                                      \
  }
  // row r is first used at k = 3r, i.e. step s = (3r)/2; fetch it one step earlier (rows 0,1 arrive via nrow)
#define constexpr_for_rows(s)                                                                            \
  _Pragma("unroll") for (int r = 2; r < 9; ++r) {                                                        \
    if ((3 * r) / 2 - 1 == (s)) load_row(row[r], lb_cur, r);                                             \
  }

  MV_C3_GROUP(accA, accB, 0, false)
  int j = 1;
  for (; j + 1 < ng; j += 2) {
    MV_C3_GROUP(accB, accA, j, true)
    MV_C3_GROUP(accA, accB, j + 1, true)
  }
  if (j < ng) {
    MV_C3_GROUP(accB, accA, j, true)
#pragma unroll
    for (int i = 0; i < 16; ++i) drain(accB, i, vo_prev, ok_prev);
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) drain(accA, i, vo_prev, ok_prev);
  }
#undef MV_C3_GROUP
#undef constexpr_for_rows
}

// ---------------------------------------------------------------------------------------------
template <bool RELU, bool FULL, bool FULLM, bool IN_U8>
static int c3_launch_t(const C3Args& a, size_t lds_bytes, hipStream_t s) {
  auto k = k_conv3x3_c3<RELU, FULL, FULLM, IN_U8>;
  if (lds_bytes > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  hipLaunchKernelGGL(k, dim3(a.nblocks), dim3(256), lds_bytes, s, a);
  return check_launch("k_conv3x3_c3");
}

template <bool RELU, bool FULL, bool FULLM>
static int c3_launch(const C3Args& a, size_t lds_bytes, bool in_u8, hipStream_t s) {
  return in_u8 ? c3_launch_t<RELU, FULL, FULLM, true>(a, lds_bytes, s) : c3_launch_t<RELU, FULL, FULLM, false>(a, lds_bytes, s);
}

bool conv3x3_c3_supported(const float* x, const float* y, int cin, int cout, int h, int w) {
  const char* v = tune_env("MV_FORCE_GENERIC_CONV");
  if (v && *v && *v != '0') return false;
  if (cin != 3 || cout > 64 || cout < 1) return false;
  if (w % 4 != 0 || (uintptr_t)y % 16 != 0) return false;  // 16-byte stores of 4 consecutive pixels
  if ((size_t)cout * h * w * sizeof(float) >= (1ull << 32)) return false;  // 32-bit per-image byte offsets
  (void)x;
  return true;
}

int launch_conv3x3_c3(const void* xv, bool in_u8, const float* mean3, const float* std3, const float* w, const float* b,
                      float* y, int64_t n, int h, int wdt, int cout, int relu, hipStream_t s) {
  C3Args a = {};
  const float* x = static_cast<const float*>(xv);
  a.x = x, a.w = w, a.b = b, a.y = y;
  for (int i = 0; i < 3; ++i) a.mean[i] = in_u8 ? mean3[i] : 0.f, a.stdv[i] = in_u8 ? std3[i] : 1.f;
  a.cout = cout, a.h = h, a.wdt = wdt, a.relu = relu;
  a.mtiles = (cout + 31) / 32;
  a.vec_rows = (wdt % 4 == 0) && ((uintptr_t)x % (in_u8 ? 4 : 16) == 0);
  const bool flat = wdt <= 256;  // whole rows per band: groups run over the flattened band
  a.wc = flat ? wdt : 256;
  a.pitch = ((a.wc + 2 + 3) & ~3) + 8;  // multiple of 4 floats, 8 floats of over-read slack per row
  int th = 16;  // 16-row bands: 14 KB contiguous per channel plane per workgroup (0.62 vs 0.68 ms at 8 rows)
  if (const char* e = tune_env("MV_C3_TH")) th = atoi(e) > 0 ? atoi(e) : th;  // tuning knob
  auto bytes = [&](int rows) {
    return ((size_t)3 * (rows + 2) * a.pitch + 16 + (size_t)a.mtiles * kKS * 64) * sizeof(float);
  };
  while (th > 1 && bytes(th) > 150 * 1024) th >>= 1;
  // small batches: a 224 x 224 image is 14 bands of 16 rows -- 14 workgroups for 256 CUs; shorter bands while the grid is small
  if (!tune_env("MV_C3_TH"))
    while (th > 4 && (long long)n * ((wdt + a.wc - 1) / a.wc) * ((h + th - 1) / th) < 512) th >>= 1;
  if (th > h) th = h;
  a.th = th;
  a.tiles_x = (wdt + a.wc - 1) / a.wc;
  a.tiles_y = (h + th - 1) / th;
  a.groups = (th * a.wc + 127) / 128;
  const long long nb = (long long)n * a.tiles_x * a.tiles_y;
  if (nb > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "conv3x3: batch too large for one launch");
  a.nblocks = (unsigned)nb;
  const size_t lds_bytes = bytes(th);
  // FULL: every lane of every group is a real pixel -> bands tile the image exactly and th*wc is a multiple of 128
  const bool full = (h % th == 0) && (wdt % a.wc == 0) && ((th * a.wc) % 128 == 0);
  const bool fullm = (cout % 32 == 0);
#define MV_C3_DISPATCH(R)                                                               \
  if (full && fullm) return c3_launch<R, true, true>(a, lds_bytes, in_u8, s);             \
  if (full) return c3_launch<R, true, false>(a, lds_bytes, in_u8, s);                     \
  if (fullm) return c3_launch<R, false, true>(a, lds_bytes, in_u8, s);                    \
  return c3_launch<R, false, false>(a, lds_bytes, in_u8, s);
  if (relu) {
    MV_C3_DISPATCH(true)
  }
  MV_C3_DISPATCH(false)
#undef MV_C3_DISPATCH
}

}  // namespace mv
