// conv3x3_gen.hip -- nn.Conv2d(cin, cout, 3, padding=1) [+bias] [+ReLU] for the deeper layers of the small CNNs
// (make_layers, models/vgg.py:73-87: 64->128->256->512 channels at 112..14 pixels): K = 9*cin >= 576, genuinely
// MFMA-bound (AI >= 288 FLOP/B at cin = 64), SURVEY.md section 8f.1.
//
// Implicit GEMM on v_mfma_f32_32x32x2_f32 (fp32 in / fp32 accumulate, bit-for-bit an ordered fmaf chain), with ONE
// accumulator per output fed in ascending k = (ci, dy, dx) order across all K-chunks and the bias as the last tap
// (A = bias, B = 1): the result equals oracle/oracle.c's chain-then-add exactly.  No split-K, no im2col buffer.
//
//   workgroup tile   128 output channels x 256 flattened pixels of one image (pixels may straddle rows: a 14x14
//                    map is one tile, a 112-wide map uses 2.3 rows per tile -- no per-row tail waste);
//   wave tile        4 channel tiles (32 each) x 2 pixel tiles (32 each): 8 independent accumulators (128 VGPRs);
//   small grids      every output is ONE chain of 9*cin dependent k-steps (64 clocks each), so a wave's time is
//                    (tiles it owns) x 9*cin/2 x 64 clk whatever the batch: 512 -> 512 channels at 14 x 14 is 0.5 ms per
//                    wave with 8 tiles, and at batch 1 that launch is 4 workgroups.  The kernel is therefore templated on
//                    the wave tile (MT channel tiles x PT pixel tiles: 4x2, 2x1, 1x1 -> workgroup tiles 128x256, 64x128,
//                    32x128) and the launcher takes the shape that minimises rounds-per-CU x tiles-per-wave;
//   K loop           chunks of 4 input channels (36 taps = 18 k-steps, fully unrolled): the chunk's zero-padded
//                    input rows and its weights (already in MFMA fragment order, one ds_read_b128 = the A operands
//                    of all 4 channel tiles) are staged in LDS; per k-step 1 + 2 LDS reads feed 8 MFMAs;
//   occupancy        __launch_bounds__(256, 2): VGPR-form MFMA, two workgroups per CU so one stages while the other
//                    computes (no intra-workgroup double buffering yet).
#include <cstdlib>
#include <type_traits>

#include "mv_common.h"
#include "mv_act.h"
#include "mv_conv.h"

#ifndef MV_GEN_TAPREGS
#define MV_GEN_TAPREGS 1  // A/B builds: 0 = the B operand's address add behind the MFMA on every tile shape
#endif
#ifndef MV_GEN_R
#define MV_GEN_R 3  // SPEC: register sets of the loader waves = chunks of global loads in flight (A/B builds: 4..6)
#endif
#ifndef MV_GEN_LEAN
#define MV_GEN_LEAN 1  // A/B builds: 0 = the loader waves stage as the general kernel does
#endif
#ifndef MV_GEN_DENSE
#define MV_GEN_DENSE 1  // lean loader with buffer loads + an LDS address set per buffer, compute loop without any VALU (A/B: 0)
#endif
#ifndef MV_GEN_ABLATE
#define MV_GEN_ABLATE 0  // profiling builds only (wrong results): 1 = no global loads after chunk 1, 2 = no LDS stores after
                         // chunk 1, 3 = neither, 4 = no MFMAs (tools/ab_conv.py)
#endif

namespace mv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // gfx950 takes 16-byte loads at any 4-byte address

constexpr int kCK = 4;                 // input channels per K-chunk
constexpr int kXP = 3 * kCK / 4;       // float4 of input a thread holds in registers per chunk
constexpr int kStepsPerChunk = kCK * 9 / 2;  // 18
// workgroup tile: 32 * MT channels x 128 * PT flattened pixels (4 waves, each MT channel tiles x PT pixel tiles)

struct GenArgs {
  const float* x;
  const float* w;
  const float* b;
  float* y;
  int cin, cout, h, wdt;
  int chunks;       // ceil(cin / 4)
  int pitch;        // LDS row pitch (floats, multiple of 4, >= wdt + 2 + 2)
  int max_rows;     // LDS rows per channel (tile rows incl. halo)
  int group;        // images stacked per "super-image" (1 = none); stacked images are separated by one zero row
  int n;            // batch size
  int pblocks;      // ceil(virtual pixels / 256) per super-image
  int cblocks;      // ceil(cout / 128)
  int relu;
  int vec_rows;     // 16-byte staging loads held in registers across a chunk: always, unless MV_CONV_NO_ROWVEC (A/B)
  int vec_w;        // weight rows 16-byte aligned (cin % 4 == 0 and aligned base)
  int ragged;       // w % 4 != 0
  int colfast;      // weight staging items: row fastest over the lanes (conflict-free LDS writes) or float4-of-a-row fastest
  unsigned nblocks;
  int slices;       // K slices across workgroups = gridDim.y (1: none)
  int cps;          // chunks per slice; == chunks without slicing
  long long slice_stride;  // floats between the outputs of consecutive slices (0 without slicing: y is the output itself)
};

// SPEC (wave specialisation, small grids): 512 threads -- waves 0-3 only read operands from LDS and issue MFMAs, waves 4-7
// only stage (global -> registers -> LDS, three chunks of loads in flight).  With one workgroup per CU and one wave per
// SIMD nothing overlapped the staging: PMC on 512 -> 512 at 28 x 28, batch 1 (profiles/r01_pmc_conv3x3_gen_batch1.txt)
// shows the matrix pipe busy 32 % of a wave's life, 33 % spent in s_waitcnt and the rest issuing the ~220 staging
// instructions per chunk.  A loader wave on the same SIMD issues those while the compute wave's MFMAs run.
// FAST: cin % 4 == 0, 16-byte weight rows and every chunk's input rows fit the register prefetch -- checked on the host, so
// the staging loads are straight-line code (the compiler can then count them: s_waitcnt vmcnt(N > 0)).
// f(integral_constant<int, I>) for I = LO .. HI-1, unrolled at compile time
template <int LO, int HI, class F>
__device__ __forceinline__ void gen_unroll(F&& f) {
  if constexpr (LO < HI) {
    f(std::integral_constant<int, LO>{});
    gen_unroll<LO + 1, HI>(f);
  }
}

template <bool RELU, int MT, int PT, bool SPEC = false, int FAST = 0>  // FAST: 0 general, 1 straight-line staging, 2 + dense (below)
__global__ __launch_bounds__(SPEC ? 512 : 256, (SPEC && MT * PT >= 8) ? 1 : 2) void k_conv3x3_gen(const GenArgs A) {
  constexpr int kBM = 32 * MT, kBP = 128 * PT;
  typedef float afrag_t __attribute__((ext_vector_type(MT)));
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool is_loader = SPEC && wave_all >= 4;             // wave-uniform role
  const int tid = SPEC ? (threadIdx.x & 255) : threadIdx.x;  // index inside the role's 256 threads
  const int lane = tid & (kWave - 1);
  const int wave = wave_all & 3;
  const int l31 = lane & 31, hf = lane >> 5;
  const int cin = A.cin, cout = A.cout, h = A.h, w = A.wdt, pitch = A.pitch;
  const int hw = h * w;
  const int Kreal = cin * 9;
  // K slice of this workgroup (blockIdx.y): chunks [cb0, cb0 + nch) -- the whole K without slicing
  const int cb0 = blockIdx.y * A.cps;
  const int nch = min(A.cps, A.chunks - cb0);

  // block -> (image, channel block, pixel block); channel blocks of one pixel block are adjacent (input reuse in L2)
  const unsigned wid = xcd_remap(blockIdx.x, A.nblocks);
  const int cb = wid % A.cblocks;
  const unsigned t2 = wid / A.cblocks;
  const int pb = t2 % A.pblocks;
  const long long sg = t2 / A.pblocks;          // super-image: images [img0, img0 + group)
  const long long img0 = sg * A.group;
  const int c0 = cb * kBM, p0 = pb * kBP;
  // Small maps (14x14, 28x28) fill a 256-pixel tile badly (196 / 256), so several images are stacked into one
  // virtual image with ONE zero separator row between them (that row is both images' padding): virtual row
  // v = g*(h+1) + y.  With group == 1 the virtual image is the image itself.
  const int hs = (A.group > 1) ? h + 1 : h;     // virtual rows per stacked image
  const int hv = A.group * hs;                  // virtual height
  const int y_first = p0 / w;
  const int y_last = min((p0 + kBP - 1) / w, hv - 1);
  const int nrows = y_last - y_first + 3;  // virtual rows y_first-1 .. y_last+1
  const int nrp = nrows * pitch;           // floats per staged channel

  constexpr int kLead = 4;  // floats in front of each buffer's rows: SPEC's unmasked row writes reach 3 floats before a row
  float* xin = lds + kLead;                      // [4][max_rows][pitch]
  float* wfr = xin + kCK * A.max_rows * pitch;   // [18][64][MT]: A operands of the MT channel tiles, fragment order
  const float* xp = A.x + (size_t)img0 * cin * hw;

  // ---- this lane's two (virtual) pixels: LDS base, validity, output byte offset relative to image img0
  int lb[PT];
  bool pvalid[PT];
  unsigned pout[PT];
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    const int pv = p0 + (PT * wave + j) * 32 + l31;
    const int pc = min(pv, hv * w - 1);   // clamp for addressing; the store is masked
    const int vy = pc / w, px = pc - vy * w;
    lb[j] = (vy - y_first) * pitch + px;  // tile row 0 <-> virtual row y_first-1, tile column 0 <-> x = -1
    const int g = vy / hs, yy = vy - g * hs;
    pvalid[j] = (pv < hv * w) && (yy < h) && (img0 + g < A.n);
    pout[j] = (unsigned)((((size_t)g * cout + 4 * hf) * hw + (size_t)yy * w + px) * sizeof(float));
  }
  const int abase = lane * MT;  // float index inside one k-step's (64 * MT)-float A slab
#ifdef MV_GEN_TRACE  // tools/trace_conv_gen.py: shader-clock stamps of workgroup 0's waves, behind the (unsliced) output tensor
  int trace_slot = 0;
  long long* const trace_buf = reinterpret_cast<long long*>(A.y + (size_t)A.n * cout * hw) + wave_all * 64;
#define MV_GEN_STAMP() do { if (blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && trace_slot < 64) trace_buf[trace_slot++] = clock64(); } while (0)
#else
#define MV_GEN_STAMP() do { } while (0)
#endif
  MV_GEN_STAMP();  // 0: start

  f32x16 acc[PT][MT];
#pragma unroll
  for (int j = 0; j < PT; ++j)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][m][i] = 0.f;

  // One pixel tile per wave (1x1, 2x1 tiles): a k-step is one or two MFMAs, and a VALU instruction that stands between an MFMA
  // and the LDS reads behind it does not issue until that MFMA has left the matrix pipe -- the B operand's address add cost the
  // wave a second MFMA time per step (tools/micro/mfma_loop.hip: 69 clocks per step without it, 140 with it, 87 with the add
  // moved in front of the MFMA; the kernel measured 113-126).  So the lane's 18 B-operand addresses of buffer 0 live in
  // registers (they do not change from chunk to chunk), the add of the buffer flip for step s + 2 is issued BEFORE the MFMA of
  // step s, in a scheduling region of its own, and the MFMA is followed by its two reads and nothing else.  (Two register sets,
  // one per buffer, and the chunk loop unrolled by two take the add away altogether -- 75 clocks per step in the compute waves --
  // but the loader wave on the same SIMD then gets too few issue slots and the workgroup waits for IT:
  // profiles/r03_trace_conv_gen.log.)
  constexpr bool kTapRegs = (PT == 1) && SPEC && MV_GEN_TAPREGS;  // (measured 0.5-1 % slower in the 256-thread kernels of large grids)
  typedef const __attribute__((address_space(3))) float lds_cf;
  constexpr bool kDense = FAST == 2 && SPEC && (MT * PT == 1) && MV_GEN_LEAN && MV_GEN_TAPREGS && MV_GEN_DENSE;
  int baddr[kTapRegs ? kStepsPerChunk : 1];  // LDS byte address of this lane's B operand at step s, buffer 0
  int baddr1[kDense ? kStepsPerChunk : 1];   // ... buffer 1 (kDense: no add left in the loop)
  if constexpr (kTapRegs) {
    const int xin0 = (int)(unsigned)(size_t)(lds + kLead);  // low half of the flat address = the LDS byte address
#pragma unroll
    for (int st = 0; st < kStepsPerChunk; ++st) {
      const int k0 = 2 * st, k1 = 2 * st + 1;
      const int o0 = (k0 / 9) * nrp + ((k0 % 9) / 3) * pitch + (k0 % 9) % 3;
      const int o1 = (k1 / 9) * nrp + ((k1 % 9) / 3) * pitch + (k1 % 9) % 3;
      baddr[st] = xin0 + 4 * (lb[0] + (hf ? o1 : o0));
      if constexpr (kDense) {
        baddr1[st] = baddr[st] + 4 * (kLead + kCK * A.max_rows * pitch + kStepsPerChunk * 64 * MT);  // + bufsz floats
        asm volatile("" : "+v"(baddr[st]));   // registers of their own: rematerialised, the adds would sit behind the MFMAs again
        asm volatile("" : "+v"(baddr1[st]));
      }
    }
  }

  // ---- chunk staging, split in two halves so that the NEXT chunk's global loads are in flight while the current
  //      chunk's MFMAs run: gload(ch) -> registers, lstore(buf) -> LDS (fragment order / zero-padded rows).
  //      Every per-thread index (source offset, LDS destination, validity) is chunk-invariant and computed once here:
  //      the loop itself contains no integer division.  LDS is double-buffered: one barrier per chunk.
  constexpr int XP = kXP;   // float4 of input per thread held in registers (maps up to ~120 wide; wider: direct staging)
  constexpr int WQ = kCK * 9 / 4;                 // float4 per channel row of a weight chunk (9)
  constexpr int WU = (kBM * WQ + 255) / 256;      // float4 of weights per thread (5 for 128 channels)
  // R register sets: chunk c's loads land in set c % R and are consumed R chunks after they were issued (SPEC: 3)
  constexpr int R = SPEC ? (kDense ? 4 : MV_GEN_R) : 1;
  f32x4 wreg[R][WU], xreg[R][XP];
  const int nq = A.vec_rows ? (((w + 4) >> 2) + 1) : (w + 2);  // groups of 4 tile columns 4q-3 .. 4q, up to column w + 1
  const int xitems = kCK * nrows * nq;
  const bool xprefetch = FAST || (A.vec_rows && xitems <= XP * 256 && w >= 4);  // (the anchored ragged-edge load needs 4 columns)
  const int bufsz = kLead + kCK * A.max_rows * pitch + kStepsPerChunk * 64 * MT;  // floats per LDS buffer (input rows + A slabs)

  long long wsrc[WU];   // element offset into A.w of this thread's float4 (chunk 0), -1 = nothing to load
  int wdst[WU];         // LDS float index of element 0 inside the A slab area
#pragma unroll
  for (int u = 0; u < WU; ++u) {
    const int idx = tid + 256 * u;
    // item -> (weight row, float4 of its chunk).  The row varies fastest over the lanes, so a wave's LDS writes spread over
    // 32 banks; with the float4 index fastest (colfast = 0, kept for A/B) the 9 float4 of a row are 128 floats apart -- one
    // bank, 9-way conflicts (SQ_LDS_BANK_CONFLICT = half of the LDS-active cycles) -- though each row is then read as one
    // contiguous 144 bytes.
    const int col = A.colfast ? idx % kBM : idx / WQ, q = A.colfast ? idx / kBM : idx - col * WQ;
    const int co = c0 + col;
    const bool witem = col < kBM && q < WQ;
    wsrc[u] = (witem && co < cout) ? (long long)co * Kreal + 4 * q : -1;
    wdst[u] = witem ? ((2 * q) * 64 + (col & 31)) * MT + (col >> 5) : -1;  // [step][half][channel & 31][channel tile]
  }
  int xsrc[XP], xdst[XP], xcil[XP];  // source offset inside a 4-channel slab (-1: zero), LDS index of element 0, channel
  unsigned xmask[XP];                // which of the 4 elements land inside the tile row
  int xsh[XP];                       // ragged right edge (w % 4 != 0): the float4 is loaded anchored at column w - 4 and
                                     // shifted left by this many elements on its way to LDS (0 = as loaded)
#pragma unroll
  for (int u = 0; u < XP; ++u) {
    const int it = tid + 256 * u;
    xsrc[u] = -1, xdst[u] = 0, xmask[u] = 0u, xcil[u] = 0, xsh[u] = 0;
    if (xprefetch && it < xitems) {
      const int r = it / nq, q = it - r * nq;
      const int cil = r / nrows, tr = r - cil * nrows;
      const int vy = y_first - 1 + tr, gx0 = 4 * q - 4;
      xcil[u] = cil;
      if (vy >= 0 && vy < hv) {
        const int g = vy / hs, gy = vy - g * hs;
        if (gy < h && img0 + g < A.n && gx0 >= 0 && gx0 < w) {
          xsrc[u] = (g * cin + cil) * hw + gy * w + gx0;
          if (gx0 + 3 >= w) xsh[u] = gx0 - (w - 4), xsrc[u] -= xsh[u];  // 1..3 valid elements: load columns w-4 .. w-1
        }
      }
      xdst[u] = cil * nrp + tr * pitch + (4 * q - 3);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 4 * q - 3 + i;
        if (c >= 0 && c < w + 2) xmask[u] |= 1u << i;
      }
    }
  }

  // FAST + SPEC (kLean): the loader wave shares its SIMD with a wave that keeps the matrix pipe busy, and a VALU instruction of
  // ANY wave on that SIMD waits for the MFMA in flight -- a chunk of staging (~60 VALU instructions: 64-bit address arithmetic,
  // dummy-load selects) took the loader 2000-2300 clocks, longer than the chunk's 18 MFMAs, and the compute waves waited at
  // the barrier (tools/trace_conv_gen.py, profiles/r03_trace_conv_gen.log).  Lean staging: the loads take a wave-uniform base
  // pointer (SALU) plus a chunk-invariant 32-bit lane offset; weight rows beyond cout are loaded as they come (their output rows are
  // never stored); input items outside the image are not written at all -- their LDS cells are zeroed once, below.
  constexpr bool kLean = FAST && SPEC && MV_GEN_LEAN;
  unsigned woff[WU], xoff[XP];
#pragma unroll
  for (int u = 0; u < WU; ++u) woff[u] = wsrc[u] >= 0 ? (unsigned)(wsrc[u] * (long long)sizeof(float)) : 0u;
#pragma unroll
  for (int u = 0; u < XP; ++u) xoff[u] = xsrc[u] >= 0 ? (unsigned)xsrc[u] * (unsigned)sizeof(float) : 0u;
  if constexpr (kLean) {
    for (int i = threadIdx.x; i < 2 * bufsz; i += 512) lds[i] = 0.f;  // both buffers, halo cells included
    __syncthreads();
  }
  // kDense: the loader's loads are buffer loads (descriptor + 32-bit lane offset + wave-uniform chunk offset: no address
  // arithmetic in the vector unit) and its LDS destinations are registers, one set per buffer
  typedef __attribute__((address_space(3))) float lds_f;
  int wda[2][kDense ? WU : 1], xda[2][kDense ? XP : 1];
  __amdgpu_buffer_rsrc_t wrs, xrs;
  if constexpr (kDense) {
    const int l0 = (int)(unsigned)(size_t)lds;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
      for (int u = 0; u < WU; ++u) {
        wda[b][u] = l0 + 4 * (b * bufsz + kLead + kCK * A.max_rows * pitch + (wdst[u] >= 0 ? wdst[u] : 0));
        asm volatile("" : "+v"(wda[b][u]));
      }
#pragma unroll
      for (int u = 0; u < XP; ++u) {
        // a ragged-edge item (loaded anchored at column w - 4) goes to LDS anchored as well: it overlaps its left neighbour's
        // cells with the same values instead of being shifted through selects; the halo column stays zero from the fill
        xda[b][u] = l0 + 4 * (b * bufsz + kLead + xdst[u] - xsh[u]);
        asm volatile("" : "+v"(xda[b][u]));
      }
    }
    wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.w), 0, -1, 0x00020000);
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xp), 0, -1, 0x00020000);
  }

  auto gload = [&](int ch, auto RC) {
    constexpr int rs = decltype(RC)::value;
    const int kbase = ch * kCK * 9;
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if constexpr (kDense) {
        v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, woff[u], kbase * (int)sizeof(float), 0));
      } else if constexpr (kLean) {
        v = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(A.w + kbase) + woff[u]);
      } else if (FAST || A.vec_w) {
        // UNCONDITIONAL load (threads without an item read the chunk's first taps and drop them in lstore): a load that
        // may be skipped forces `s_waitcnt vmcnt(0)` on every later use of ANY earlier load -- the counter only says how
        // many loads are outstanding -- and that serialised the prefetch: one chunk in flight instead of R.
        v = *reinterpret_cast<const f32x4*>(A.w + (wsrc[u] >= 0 ? wsrc[u] : 0) + kbase);
      } else if (wsrc[u] >= 0) {
        const float* src = A.w + wsrc[u] + kbase;
        {
          const int kq = (int)(wsrc[u] % Kreal) + kbase;  // tap index of element 0 (slow path: cin % 4 != 0)
          if (kq + 0 < Kreal) v.x = src[0];
          if (kq + 1 < Kreal) v.y = src[1];
          if (kq + 2 < Kreal) v.z = src[2];
          if (kq + 3 < Kreal) v.w = src[3];
        }
      }
      wreg[rs][u] = v;
    }
    if (FAST || xprefetch) {
      const float* slab = xp + (size_t)ch * kCK * hw;
#pragma unroll
      for (int u = 0; u < XP; ++u) {
        // one kind of load on every path, and unconditional (see the weights above); rows / columns / channels outside the
        // image read the slab's first pixels and are zeroed in lstore
        if constexpr (kDense) {
          xreg[rs][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff[u], (int)((unsigned)(ch * kCK) * (unsigned)hw * 4u), 0));
        } else if constexpr (kLean) {  // cin % 4 == 0: every channel of the chunk exists
          xreg[rs][u] = *reinterpret_cast<const f32x4u*>(reinterpret_cast<const char*>(slab) + xoff[u]);
        } else {
          const bool ok = xsrc[u] >= 0 && ch * kCK + xcil[u] < cin;
          xreg[rs][u] = *reinterpret_cast<const f32x4u*>(slab + (ok ? xsrc[u] : 0));
        }
      }
    }
  };
  auto lstore = [&](int ch, float* xin_b, float* wfr_b, auto RC) {
    constexpr int rs = decltype(RC)::value;
    if constexpr (kDense) {  // chunk k is staged from register set k % 4 into buffer k & 1 = rs & 1
      constexpr int par = rs & 1;
#pragma unroll
      for (int u = 0; u < WU; ++u) {
        if (wdst[u] >= 0) {
          lds_f* d = reinterpret_cast<lds_f*>((size_t)(unsigned)wda[par][u]);
          d[0] = wreg[rs][u].x, d[32 * MT] = wreg[rs][u].y, d[64 * MT] = wreg[rs][u].z, d[96 * MT] = wreg[rs][u].w;
        }
      }
#pragma unroll
      for (int u = 0; u < XP; ++u) {
        if (xsrc[u] < 0 || xmask[u] == 0u) continue;  // outside the image: the cells keep their zeros
        lds_f* d = reinterpret_cast<lds_f*>((size_t)(unsigned)xda[par][u]);
        d[0] = xreg[rs][u].x, d[1] = xreg[rs][u].y, d[2] = xreg[rs][u].z, d[3] = xreg[rs][u].w;
      }
      return;
    }
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      if (wdst[u] >= 0) {
        float* d = wfr_b + wdst[u];
        const bool wv = kLean || wsrc[u] >= 0 || !(FAST || A.vec_w);  // channel rows beyond cout hold zeros (vec_w: loaded a dummy)
        d[0] = wv ? wreg[rs][u].x : 0.f, d[32 * MT] = wv ? wreg[rs][u].y : 0.f;  // half -> +32 lanes, step -> +64
        d[64 * MT] = wv ? wreg[rs][u].z : 0.f, d[96 * MT] = wv ? wreg[rs][u].w : 0.f;
      }
    }
    if (FAST || xprefetch) {
#pragma unroll
      for (int u = 0; u < XP; ++u) {
        float* d = xin_b + xdst[u];
        if constexpr (kLean) {
          if (xsrc[u] < 0) continue;  // outside the image: the cells keep the zeros they were given at the start
        } else {
          if (!(xsrc[u] >= 0 && ch * kCK + xcil[u] < cin)) xreg[rs][u] = (f32x4){0.f, 0.f, 0.f, 0.f};  // the dummy load
        }
        if (A.ragged) {  // wave-uniform: w % 4 != 0
          const f32x4 a = xreg[rs][u];
          const int sh = xsh[u];
          xreg[rs][u].x = sh == 0 ? a.x : (sh == 1 ? a.y : (sh == 2 ? a.z : a.w));
          xreg[rs][u].y = sh == 0 ? a.y : (sh == 1 ? a.z : (sh == 2 ? a.w : 0.f));
          xreg[rs][u].z = sh == 0 ? a.z : (sh == 1 ? a.w : 0.f);
          xreg[rs][u].w = sh == 0 ? a.w : 0.f;
        }
        if constexpr (SPEC) {
          // unmasked: elements left of tile column 0 fall into the previous row's padding (pitch >= w + 6) or the kLead
          // floats in front of the buffer, elements right of column w + 1 into this row's padding; nobody reads padding
          if (xmask[u] != 0u) d[0] = xreg[rs][u].x, d[1] = xreg[rs][u].y, d[2] = xreg[rs][u].z, d[3] = xreg[rs][u].w;
        } else {
          if (xmask[u] & 1u) d[0] = xreg[rs][u].x;
          if (xmask[u] & 2u) d[1] = xreg[rs][u].y;
          if (xmask[u] & 4u) d[2] = xreg[rs][u].z;
          if (xmask[u] & 8u) d[3] = xreg[rs][u].w;
        }
      }
    } else {
      // direct staging (wide or unaligned maps): tile column c <-> gx = c - 1
      const int cols = w + 2;
      const int nitems = kCK * nrows * cols;
      for (int it = tid; it < nitems; it += 256) {
        const int r = it / cols, c = it - r * cols;
        const int cil = r / nrows, tr = r - cil * nrows;
        const int ci = ch * kCK + cil, vy = y_first - 1 + tr, gx = c - 1;
        float v = 0.f;
        if (ci < cin && vy >= 0 && vy < hv && gx >= 0 && gx < w) {
          const int g = vy / hs, gy = vy - g * hs;
          if (gy < h && img0 + g < A.n) v = xp[((size_t)g * cin + ci) * hw + (size_t)gy * w + gx];
        }
        xin_b[cil * nrp + tr * pitch + c] = v;
      }
    }
  };

  using ic0 = std::integral_constant<int, 0>;
  if constexpr (SPEC) {
    if (is_loader) {
      // ---- loader waves: one barrier per chunk, in step with the compute waves below
      // every gload below is executed whatever the chunk count (indices clamped to the last chunk: a few redundant loads at
      // the tail): a load that may be skipped cannot be counted on by s_waitcnt vmcnt(N)
      const int last = nch - 1;
      gload(cb0, ic0{});
      lstore(cb0, xin, wfr, ic0{});
      __builtin_amdgcn_sched_barrier(0);  // issue order = consumption order, or the loop's waits degrade to vmcnt(0)
      gen_unroll<1, R>([&](auto I) {      // chunks 1 .. R-1 into sets 1 .. R-1
        gload(cb0 + min((int)decltype(I)::value, last), I);
        __builtin_amdgcn_sched_barrier(0);
      });
      gload(cb0 + min(R, last), ic0{});
      __syncthreads();
      // one step per chunk: chunk c + 1 goes registers -> LDS (the buffer nobody reads during chunk c), its register set is
      // refilled with chunk c + 1 + R, barrier.  Unrolled by the ring length with the sets named statically: the compiler's
      // s_waitcnt insertion then sees which loads are older than the ones a step consumes and waits with vmcnt(5 (R - 1)), not
      // vmcnt(0) -- selected through a run-time index it waited for everything, i.e. prefetched one chunk ahead, not R.
      auto step = [&](int c, auto RC) {
        MV_GEN_STAMP();  // loader +0: past the barrier
        if (c + 1 < nch && (!(MV_GEN_ABLATE & 2) || c < 1)) {
          float* xin_n = lds + ((c + 1) & 1) * bufsz + kLead;
          float* wfr_n = xin_n + kCK * A.max_rows * pitch;
          lstore(cb0 + c + 1, xin_n, wfr_n, RC);
        }
        MV_GEN_STAMP();  // loader +1: chunk c + 1 written to LDS (issued)
        if (!(MV_GEN_ABLATE & 1) || c < 1) gload(cb0 + min(c + 1 + R, last), RC);
        MV_GEN_STAMP();  // loader +2: chunk c + 1 + R requested
        __syncthreads();
      };
      for (int ch = 0; ch < nch; ch += R)  // the compute waves pad their barrier count to a multiple of R as well
        gen_unroll<0, R>([&](auto I) { step(ch + (int)decltype(I)::value, std::integral_constant<int, (decltype(I)::value + 1) % R>{}); });
      return;
    }
    __syncthreads();  // chunk 0 staged by the loaders
  } else {
    gload(cb0, ic0{});
    lstore(cb0, xin, wfr, ic0{});
    if (nch > 1) gload(cb0 + 1, ic0{});
    __syncthreads();
  }
  for (int ch = 0; ch < nch; ++ch) {
    MV_GEN_STAMP();  // compute +0: past the barrier
    float* xin_c = lds + (ch & 1) * bufsz + kLead;
    float* wfr_c = xin_c + kCK * A.max_rows * pitch;
    float* xin_n = lds + ((ch + 1) & 1) * bufsz + kLead;
    float* wfr_n = xin_n + kCK * A.max_rows * pitch;
    const float* xin = xin_c;  // shadow the outer names for the k-step code below
    const float* wfr = wfr_c;

    // ---- 36 k-steps, fully unrolled: k = 2s + hf inside the chunk -> (channel, dy, dx).  Operands of step s+1 are
    //      fetched before the MFMAs of step s; a scheduling barrier per step keeps the compiler from hoisting all
    //      108 LDS reads to the top (which spilled the accumulators).
    auto fetch = [&](int s, float (&av)[MT], float (&bv)[PT]) {
      const int k0 = 2 * s, k1 = 2 * s + 1;
      const int o0 = (k0 / 9) * nrp + ((k0 % 9) / 3) * pitch + (k0 % 9) % 3;  // wave-uniform
      const int o1 = (k1 / 9) * nrp + ((k1 % 9) / 3) * pitch + (k1 % 9) % 3;
      const int o = hf ? o1 : o0;
      if constexpr (MT == 1) {
        av[0] = wfr[s * 64 + abase];
      } else {
        const afrag_t t = *reinterpret_cast<const afrag_t*>(wfr + s * (64 * MT) + abase);  // one ds_read_b64 / b128
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = t[m];
      }
#pragma unroll
      for (int j = 0; j < PT; ++j) bv[j] = xin[lb[j] + o];
    };
    if constexpr (!kTapRegs) {
      float av_c[MT], av_n[MT], bv_c[PT], bv_n[PT];
      fetch(0, av_c, bv_c);
#pragma unroll
      for (int s = 0; s < kStepsPerChunk; ++s) {
        if (s + 1 < kStepsPerChunk) fetch(s + 1, av_n, bv_n);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
          for (int j = 0; j < PT; ++j) {
            if (MV_GEN_ABLATE == 4) acc[j][m][s & 15] += av_c[m] * bv_c[j];
            else acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av_c[m], bv_c[j], acc[j][m], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) av_c[m] = av_n[m];
#pragma unroll
        for (int j = 0; j < PT; ++j) bv_c[j] = bv_n[j];
      }
    } else {
      auto fetch_a = [&](int s, float (&av)[MT]) {
        if constexpr (MT == 1) {
          av[0] = wfr[s * 64 + abase];
        } else {
          const afrag_t t = *reinterpret_cast<const afrag_t*>(wfr + s * (64 * MT) + abase);
#pragma unroll
          for (int m = 0; m < MT; ++m) av[m] = t[m];
        }
      };
      if constexpr (kDense) {
        auto steps = [&](const int (&ba)[kStepsPerChunk]) {
          float avr[2][MT], bvr[2];
          fetch_a(0, avr[0]);
          bvr[0] = *reinterpret_cast<lds_cf*>((size_t)(unsigned)ba[0]);
#pragma unroll
          for (int s = 0; s < kStepsPerChunk; ++s) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
              acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(avr[s & 1][m], bvr[s & 1], acc[0][m], 0, 0, 0);
            if (s + 1 < kStepsPerChunk) {
              fetch_a(s + 1, avr[(s + 1) & 1]);
              bvr[(s + 1) & 1] = *reinterpret_cast<lds_cf*>((size_t)(unsigned)ba[s + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        if (ch & 1) steps(baddr1);
        else steps(baddr);
      } else {
      const int flipb = (ch & 1) * bufsz * (int)sizeof(float);  // wave-uniform
      int an[3];  // B-operand addresses of steps s, s + 1, s + 2
      float avr[2][MT], bvr[2];
      an[0] = baddr[0] + flipb, an[1] = baddr[1] + flipb;
      fetch_a(0, avr[0]);
      bvr[0] = *reinterpret_cast<lds_cf*>((size_t)(unsigned)an[0]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < kStepsPerChunk; ++s) {
        if (s + 2 < kStepsPerChunk) {
          an[(s + 2) % 3] = baddr[s + 2] + flipb;
          asm volatile("" : "+v"(an[(s + 2) % 3]));  // the add stays in this region, in front of the MFMA
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if (MV_GEN_ABLATE == 4) acc[0][m][s & 15] += avr[s & 1][m] * bvr[s & 1];
          else acc[0][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(avr[s & 1][m], bvr[s & 1], acc[0][m], 0, 0, 0);
        }
        if (s + 1 < kStepsPerChunk) {
          fetch_a(s + 1, avr[(s + 1) & 1]);
          bvr[(s + 1) & 1] = *reinterpret_cast<lds_cf*>((size_t)(unsigned)an[(s + 1) % 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      }
    }
    MV_GEN_STAMP();  // compute +1: the chunk's MFMAs issued
    if (!SPEC && ch + 1 < nch) {
      if (!(MV_GEN_ABLATE & 2) || ch < 1) lstore(cb0 + ch + 1, xin_n, wfr_n, ic0{});  // the other buffer: nobody reads it during this chunk
      if (ch + 2 < nch && (!(MV_GEN_ABLATE & 1) || ch < 1)) gload(cb0 + ch + 2, ic0{});  // in flight during the next chunk's MFMAs
    }
    __syncthreads();
  }

  if constexpr (SPEC) {  // the loaders run whole rounds of R steps
    for (int c = nch; c % R != 0; ++c) __syncthreads();
  }

  // ---- bias as the last tap: A = bias[channel] on the k-even half, B = 1 there and 0 on the odd half
  if (A.b != nullptr) {
    const float bsel = hf ? 0.f : 1.f;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int co = c0 + 32 * m + l31;
      const float av = (hf == 0 && co < cout) ? A.b[co] : 0.f;
#pragma unroll
      for (int j = 0; j < PT; ++j) acc[j][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bsel, acc[j][m], 0, 0, 0);
    }
  }

  // ---- ReLU + store: register i of tile (j, m) is channel c0 + 32m + (i&3) + 8(i>>2) + 4hf at this lane's pixel
  char* const simg = reinterpret_cast<char*>(A.y + (size_t)blockIdx.y * A.slice_stride + (size_t)img0 * cout * hw);
#pragma unroll
  for (int j = 0; j < PT; ++j) {
    if (pvalid[j]) {
      const unsigned voff = pout[j];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int cu = c0 + 32 * m + (i & 3) + 8 * (i >> 2);  // + 4*hf
          float v = acc[j][m][i];
          if (RELU) v = relu_f32(v);
          if (cu + 4 * hf < cout) *reinterpret_cast<float*>(simg + (size_t)cu * hw * sizeof(float) + voff) = v;
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
bool conv3x3_gen_supported(int cin, int cout, int h, int w) {
  const char* v = tune_env("MV_FORCE_SIMPLE_CONV");
  if (v && *v && *v != '0') return false;
  if (cin < 1 || cout < 1 || w > 510) return false;  // whole rows must fit the LDS tile
  if ((size_t)(cout + 4) * h * w * sizeof(float) >= (1ull << 32)) return false;  // 32-bit per-image byte offsets
  return true;
}

// Workgroups of the (MT, PT) shape for this problem, with the image stacking it would use (*group_out)
static long long gen_grid(int64_t n, int h, int wdt, int cout, int mt, int pt, int* group_out) {
  const int bm = 32 * mt, bp = 128 * pt;
  // Stack images of small maps when that lowers the number of pixel tiles -- but only while the grid stays
  // >= 3 workgroups per CU: with fewer, every workgroup runs concurrently anyway and a smaller grid buys nothing
  // (measured: batch 64 at 14x14 got slower with stacking, 272 workgroups on 256 CUs).
  int group = 1;
  const int cblocks = (cout + bm - 1) / bm;
  long long best = (long long)n * ((h * wdt + bp - 1) / bp);
  for (int g = 2; g <= 64 && g <= n; g *= 2) {
    if ((size_t)g * (cout + 4) * h * wdt * sizeof(float) >= (1ull << 32)) break;
    const long long tiles = ((n + g - 1) / g) * (((long long)g * (h + 1) * wdt + bp - 1) / bp);
    if (tiles < best && tiles * cblocks >= 768) best = tiles, group = g;
  }
  if (const char* e = tune_env("MV_CONV_GROUP")) group = atoi(e) > 0 ? atoi(e) : group;
  if (group > n) group = (int)n;
  *group_out = group;
  const int hv = group > 1 ? group * (h + 1) : h;
  const long long pblocks = ((long long)hv * wdt + bp - 1) / bp;
  return ((n + group - 1) / group) * pblocks * cblocks;
}

template <int MT, int PT, bool SPEC = false>
static int launch_gen_shape(GenArgs& a, int64_t n, int group, hipStream_t s) {
  constexpr int kBM = 32 * MT, kBP = 128 * PT;
  const int h = a.h, wdt = a.wdt, cout = a.cout;
  a.group = group;
  if (a.colfast < 0) a.colfast = 1;  // measured 2-8 % faster on every shape and batch (profiles/r01_tune_conv_gen_colfast.log)
  const int hv = group > 1 ? group * (h + 1) : h;
  const int span = (kBP + wdt - 1) / wdt + 1;  // rows a kBP-pixel run can touch
  a.max_rows = (span < hv ? span : hv) + 2;
  a.pblocks = (hv * wdt + kBP - 1) / kBP;
  a.cblocks = (cout + kBM - 1) / kBM;
  const long long nsuper = (n + group - 1) / group;
  const long long nb = nsuper * a.pblocks * a.cblocks;
  if (nb > 0x7fffffffLL) return set_error(MV_ERR_UNSUPPORTED, "conv3x3: batch too large for one launch");
  a.nblocks = (unsigned)nb;
  const size_t lds_bytes = 2 * (4 + (size_t)kCK * a.max_rows * a.pitch + (size_t)kStepsPerChunk * 64 * MT) * sizeof(float);
  if (lds_bytes > 160 * 1024)
    return set_error(MV_ERR_UNSUPPORTED, "conv3x3: %dx%d feature map needs %zu B of LDS per workgroup", h, wdt, lds_bytes);
  auto launch = [&](auto kern) {
    if (lds_bytes > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(kern, dim3(a.nblocks, (unsigned)a.slices), dim3(SPEC ? 512 : 256), lds_bytes, s, a);
    return check_launch("k_conv3x3_gen");
  };
  // FAST: the kernel's `xprefetch` and `vec_w` conditions hold for every workgroup (XP = 3 float4 per thread, rows of
  // nq groups of 4 columns), so it is compiled without the other staging paths
  const int nq = ((wdt + 4) >> 2) + 1;
  bool fast = a.vec_w && a.vec_rows && wdt >= 4 && (long long)kCK * a.max_rows * nq <= kXP * 256 &&
              (long long)cout * a.cin * 9 * 4 < (1LL << 32) && (long long)group * a.cin * h * wdt * 4 < (1LL << 32);  // 32-bit lane offsets
  if (const char* e = tune_env("MV_CONV_FAST")) fast = fast && atoi(e) != 0;  // tuning knob: 0 = the general kernel
  if (fast) {
    // dense: the 1x1 tile with loader waves (profiles/r03_ab_conv_dense.log)
    bool dense = SPEC && MT * PT == 1;
    if (const char* e = tune_env("MV_CONV_DENSE")) dense = dense && atoi(e) != 0;  // tuning knob
    if constexpr (SPEC && MT * PT == 1)
      if (dense) return a.relu ? launch(k_conv3x3_gen<true, MT, PT, SPEC, 2>) : launch(k_conv3x3_gen<false, MT, PT, SPEC, 2>);
    return a.relu ? launch(k_conv3x3_gen<true, MT, PT, SPEC, 1>) : launch(k_conv3x3_gen<false, MT, PT, SPEC, 1>);
  }
  return a.relu ? launch(k_conv3x3_gen<true, MT, PT, SPEC, 0>) : launch(k_conv3x3_gen<false, MT, PT, SPEC, 0>);
}

// Wave-tile shape: a wave's time is (tiles it owns) x the K chain, a CU's time that times the workgroups it is dealt
// (256 CUs; the matrix pipe is shared by the workgroups resident on a CU), so take the shape that minimises
// ceil(workgroups / 256) x MT x PT; smaller tiles stage the same inputs for more channel blocks (+10 % / +20 %).
static long long gen_pick_shape(int64_t n, int h, int wdt, int cout, int* pick_out, int* group_out) {
  static const int shapes[3][2] = {{4, 2}, {2, 1}, {1, 1}};
  static const double overhead[3] = {1.0, 1.1, 1.2};
  int pick = 0, group = 1;
  long long wgs_pick = 0;
  double best = 0.0;
  for (int i = 0; i < 3; ++i) {
    int g;
    const long long wgs = gen_grid(n, h, wdt, cout, shapes[i][0], shapes[i][1], &g);
    const double cost = (double)((wgs + 255) / 256) * shapes[i][0] * shapes[i][1] * overhead[i];
    if (i == 0 || cost < best) best = cost, pick = i, group = g, wgs_pick = wgs;
  }
  if (const char* e = tune_env("MV_CONV_SHAPE")) {  // tuning knob: 0 = 4x2, 1 = 2x1, 2 = 1x1
    const int v = atoi(e);
    if (v >= 0 && v < 3) pick = v, wgs_pick = gen_grid(n, h, wdt, cout, shapes[v][0], shapes[v][1], &group);
  }
  *pick_out = pick, *group_out = group;
  return wgs_pick;
}

// K slices ACROSS workgroups (mv_conv3x3_bias_relu_ws_f32): a launch of a few dozen workgroups that each walk hundreds of K
// chunks in sequence (VGG's 512 -> 512 layers at batch 1: ~50 workgroups x 128 chunks, 117 us for 0.9 GFLOP) is cut into `slices`
// launches' worth of workgroups over channel ranges of `slice_channels`; their raw sums go to a workspace and
// k_conv3x3_reduce adds them in ascending slice order, then bias, then ReLU.  One chain per slice from +0 in (channel, ky, kx)
// order -- the order the oracle restates (orc_conv3x3_sliced_bias_relu_f32).
void conv3x3_gen_plan(int64_t n, int cin, int h, int wdt, int cout, int* slices, int* slice_channels) {
  *slices = 1, *slice_channels = cin;
  if (n <= 0 || h <= 0 || wdt <= 0 || !conv3x3_gen_supported(cin, cout, h, wdt)) return;
  int pick, group;
  const long long wgs = gen_pick_shape(n, h, wdt, cout, &pick, &group);
  const int chunks = (cin + kCK - 1) / kCK;
  int sl = 1;
  while (sl < 8 && wgs * (sl * 2) <= 320 && chunks / (sl * 2) >= 16) sl *= 2;
  if (const char* e = tune_env("MV_CONV_KSLICES")) sl = atoi(e) >= 1 ? atoi(e) : sl;
  if (sl <= 1) return;
  const int cps = (chunks + sl - 1) / sl;
  *slices = (chunks + cps - 1) / cps;
  *slice_channels = cps * kCK;
  if (*slices <= 1) *slices = 1, *slice_channels = cin;
}

int64_t conv3x3_gen_workspace_bytes(int64_t n, int cin, int h, int wdt, int cout) {
  int sl, sc;
  conv3x3_gen_plan(n, cin, h, wdt, cout, &sl, &sc);
  return sl > 1 ? (int64_t)sizeof(float) * sl * n * cout * h * wdt : 0;
}

struct GenReduceArgs {
  const float* part;
  const float* b;
  float* y;
  long long total, slice_stride;
  int slices, cout, hw, relu;
};

__global__ __launch_bounds__(256) void k_conv3x3_reduce(const GenReduceArgs A) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= A.total) return;
  float v = A.part[i];
  for (int s = 1; s < A.slices; ++s) v = v + A.part[(size_t)s * A.slice_stride + i];
  if (A.b != nullptr) v = v + A.b[(i / A.hw) % A.cout];
  if (A.relu) v = relu_f32(v);
  A.y[i] = v;
}

int launch_conv3x3_gen_ws(const float* x, const float* w, const float* b, float* y, int64_t n, int cin, int h, int wdt,
                          int cout, int relu, hipStream_t s, void* workspace, int64_t workspace_bytes) {
  GenArgs a = {};
  a.x = x, a.w = w, a.b = b, a.y = y;
  a.cin = cin, a.cout = cout, a.h = h, a.wdt = wdt, a.relu = relu;
  a.chunks = (cin + kCK - 1) / kCK;
  a.slices = 1, a.cps = a.chunks, a.slice_stride = 0;
  a.colfast = -1;
  a.ragged = (wdt % 4 != 0);
  if (const char* e = tune_env("MV_CONV_COLFAST")) a.colfast = atoi(e) != 0;  // tuning knob
  a.pitch = ((wdt + 2 + 3) & ~3) + 4;
  a.n = (int)n;
  a.vec_rows = 1;
  if (const char* e = tune_env("MV_CONV_NO_ROWVEC")) a.vec_rows = !(*e && *e != '0');
  a.vec_w = (cin % kCK == 0) && ((uintptr_t)w % 16 == 0);
  int pick, group;
  (void)gen_pick_shape(n, h, wdt, cout, &pick, &group);
  // K slices across workgroups: only through the entry point that brings a workspace (the plain one keeps the single chain)
  GenReduceArgs r = {};
  if (workspace != nullptr) {
    int sl, sc;
    conv3x3_gen_plan(n, cin, h, wdt, cout, &sl, &sc);
    if (sl > 1) {
      const int64_t need = (int64_t)sizeof(float) * sl * n * cout * h * wdt;
      if (workspace_bytes < need)
        return set_error(MV_ERR_INVALID_ARGUMENT, "conv3x3: workspace of %lld bytes needed (mv_conv3x3_workspace_bytes), got %lld",
                         (long long)need, (long long)workspace_bytes);
      a.slices = sl, a.cps = sc / kCK, a.slice_stride = (long long)n * cout * h * wdt;
      a.y = static_cast<float*>(workspace), a.b = nullptr, a.relu = 0;
      r.part = a.y, r.b = b, r.y = y, r.total = a.slice_stride, r.slice_stride = a.slice_stride;
      r.slices = sl, r.cout = cout, r.hw = h * wdt, r.relu = relu;
    }
  }
  int rc;
  if (pick == 0) {
    // at most one workgroup per CU: nothing else overlaps the staging, so the same loader / compute specialisation (205 VGPRs:
    // one 512-thread workgroup per CU).  512 -> 512 at 14 x 14, batch 64 (256 workgroups): 686 -> 636 us
    // (profiles/r03_ab_conv_shapes_batch64.log); slower on every larger grid, where two workgroups per CU overlap each other
    int g0;
    bool spec42 = gen_grid(n, h, wdt, cout, 4, 2, &g0) <= 256;
    if (const char* e = tune_env("MV_CONV_SPEC42")) spec42 = atoi(e) != 0;  // tuning knob
    rc = spec42 ? launch_gen_shape<4, 2, true>(a, n, group, s) : launch_gen_shape<4, 2>(a, n, group, s);
  } else if (pick == 1) {
    int g1;
    bool spec21 = gen_grid(n, h, wdt, cout, 2, 1, &g1) <= 512;  // at most ~2 workgroups per CU: the same specialisation
    if (const char* e = tune_env("MV_CONV_SPEC21")) spec21 = atoi(e) != 0;  // tuning knob
    rc = spec21 ? launch_gen_shape<2, 1, true>(a, n, group, s) : launch_gen_shape<2, 1>(a, n, group, s);
  } else {
    // one-tile-per-wave shape on a grid of at most ~2 workgroups per CU: loader / compute wave specialisation
    int g2;
    bool spec = gen_grid(n, h, wdt, cout, 1, 1, &g2) <= 512;
    if (const char* e = tune_env("MV_CONV_SPEC")) spec = atoi(e) != 0;  // tuning knob
    rc = spec ? launch_gen_shape<1, 1, true>(a, n, group, s) : launch_gen_shape<1, 1>(a, n, group, s);
  }
  if (rc != MV_OK || a.slices == 1) return rc;
  hipLaunchKernelGGL(k_conv3x3_reduce, dim3((unsigned)((r.total + 255) / 256)), dim3(256), 0, s, r);
  return check_launchf("k_conv3x3_gen + k_conv3x3_reduce<ks%d>", a.slices);
}

int launch_conv3x3_gen(const float* x, const float* w, const float* b, float* y, int64_t n, int cin, int h, int wdt,
                       int cout, int relu, hipStream_t s) {
  return launch_conv3x3_gen_ws(x, w, b, y, n, cin, h, wdt, cout, relu, s, nullptr, 0);  // no workspace: one chain per output
}

}  // namespace mv
