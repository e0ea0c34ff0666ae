/*
 * mi355vision.h -- C ABI of libmi355vision.so: the MI355X (gfx950) implementation of the
 * reference's data-parallel filtering / first-conv hot path.
 *
 * This is the drop-in boundary.  The reference (a torchvision 0.20 fork) is Python; its
 * "FFI" for this path is the pair of torch calls every filter kernel makes,
 *     torch.nn.functional.pad(x, [kx//2, kx//2, ky//2, ky//2], mode="reflect")
 *     torch.nn.functional.conv2d(x, kernel.expand(C,1,ky,kx), groups=C)
 * (transforms/v2/functional/_misc.py:153-155, _color.py:260,
 *  transforms/_functional_tensor.py:759-761, 818) and nn.Conv2d(3,64,3,padding=1)+ReLU
 * (models/vgg.py:81-85).  Each entry point below replaces one such call site (cited per
 * function, paths relative to the reference root).  The reference-side binding a maintainer
 * would add is the ctypes stub shown in INTEGRATION.md; cpu-vision_amd/_lib.py is that stub.
 *
 * Conventions (all entry points):
 *   - plain C: pointers + sizes, no torch / C++ types.
 *   - x, y, ... are DEVICE pointers owned by the caller; the library never allocates, frees or
 *     retains them.  Outputs must not alias inputs.
 *   - images are planar, contiguous: `planes` = product of all leading dims (N*C), each plane
 *     H x W, W fastest (the reference's NCHW layout).
 *   - filter taps are HOST pointers (a handful of floats, passed to the kernel by value),
 *     unless a parameter says otherwise.
 *   - `stream` is a hipStream_t (NULL = the null stream).  Calls are asynchronous with respect
 *     to the host, exactly like a torch op on that stream.
 *   - return 0 on success, a negative mv_status otherwise; mv_last_error() gives the message
 *     for the calling thread.  Nothing throws or aborts across this boundary.
 *   - re-entrant; no global mutable state besides the thread-local error / last-kernel strings, and no
 *     environment lookups: kernel selection depends on the arguments alone.
 *   - numerics: fp32 taps and accumulation, one fused multiply-add chain per output in
 *     row-major tap order starting from +0 (bit-identical to oracle/oracle.c); uint8 paths
 *     convert to fp32, accumulate, round half-to-even (torch.round_) and narrow.
 */
#ifndef MI355VISION_H
#define MI355VISION_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MV_ABI_VERSION 1

typedef enum mv_status {
  MV_OK = 0,
  MV_ERR_INVALID_ARGUMENT = -1, /* shapes / sizes / enum values the reference would also reject   */
  MV_ERR_UNSUPPORTED = -2,      /* valid request outside what the kernels cover (message says what) */
  MV_ERR_LAUNCH = -3,           /* HIP reported an error at launch                                 */
  MV_ERR_NO_DEVICE = -4         /* no gfx950 device visible                                        */
} mv_status;

typedef enum mv_border {
  MV_BORDER_VALID = 0,   /* no padding: output (H-ky+1) x (W-kx+1)      -- _color.py:260          */
  MV_BORDER_REFLECT = 1, /* pad(mode="reflect") then conv: output H x W  -- _misc.py:153-155       */
  MV_BORDER_ZERO = 2     /* zero padding k//2: output H x W              -- nn.Conv2d(padding=k//2) */
} mv_border;

/* Largest 2-D tap count that may be passed from a host pointer (by-value kernel argument). */
#define MV_MAX_HOST_TAPS_2D 121
/* Largest 1-D kernel size of the Gaussian / separable entry points. */
#define MV_MAX_TAPS_1D 63

int mv_abi_version(void);
const char* mv_last_error(void);
/* Name of the kernel instantiation the calling thread's last entry point launched (e.g.
 * "k_dwtile<f32,3x3,rpt4,vec16,tw256>"): what a benchmark reports next to its roofline numbers, taken from the
 * launcher that made the choice instead of being assumed by the caller.  "" before the first launch. */
const char* mv_last_kernel(void);
/* Identifies the build: first 16 hex digits of the SHA-256 over the library's sources and compile flags
 * ("+tuning" appended for a -DMV_TUNING build, the only kind that reads MV_* environment knobs). */
const char* mv_build_id(void);
/* Number of visible HIP devices (does not create a context). */
int mv_device_count(void);

/* ---- the primitive ------------------------------------------------------------------------
 * y = conv2d(pad(x, border), w.expand(C,1,ky,kx), groups=C): depthwise cross-correlation with one
 * (ky,kx) kernel shared by all planes.  Replaces the pad+conv2d pair at _misc.py:153-155 /
 * _functional_tensor.py:759-761 (REFLECT) and the bare conv2d at _color.py:260 (VALID).
 * `w`: ky*kx floats, row-major; host pointer when w_on_device == 0 (ky*kx <= MV_MAX_HOST_TAPS_2D),
 * device pointer otherwise.  REFLECT requires ky/2 < H and kx/2 < W (as ATen does). */
int mv_depthwise_conv2d_f32(const float* x, float* y, const float* w, int w_on_device, int64_t planes,
                            int h, int wdt, int ky, int kx, int border, void* stream);
/* Same with uint8 storage: x.to(float32) -> conv -> round_() -> .to(uint8)
 * (_misc.py:150, 160-161; _functional_tensor.py:516-542). */
int mv_depthwise_conv2d_u8(const uint8_t* x, uint8_t* y, const float* w, int w_on_device, int64_t planes,
                           int h, int wdt, int ky, int kx, int border, void* stream);

/* ---- gaussian_blur_image core (_misc.py:147-155, v1 _functional_tensor.py:746-764) -----------
 * kernel2d[j][i] = k1d_y[j] * k1d_x[i] (formed in-kernel, bit-identical to _misc.py:97), reflect
 * border, one 2-D pass exactly like the reference.  k1d_* are host pointers, kx, ky odd,
 * <= MV_MAX_TAPS_1D. */
int mv_gaussian_blur_f32(const float* x, float* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx,
                         const float* k1d_y, int ky, void* stream);
int mv_gaussian_blur_u8(const uint8_t* x, uint8_t* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx,
                        const float* k1d_y, int ky, void* stream);
/* mv_gaussian_blur_u8's result -- the reference's single 2-D pass: .to(float32) -> conv2d(outer-product kernel) -> round_() --
 * BIT FOR BIT, at the cost of the separable pair.  The separable sum is another association of the 2-D chain; the two differ by
 * at most M = (kx*ky + kx + ky + 2) * 2^-17 on uint8 data with Gaussian taps, so their roundings agree wherever the value is
 * farther than M from a tie n + 0.5.  Pass 1 (the separable kernel) stores every pixel and lists the lane-rows (16 / 4 / 2
 * pixels) that hold a value within M of a tie -- 1-4 % of them -- in `workspace`; pass 2 recomputes those pixels with the 2-D
 * chain (csrc/tiefix_u8.hip has the bound's derivation; a list that overflows makes pass 2 recompute every pixel).
 * 32 x 4K uint8: 5x5 0.64 -> 0.59 ms, 7x7 1.2 -> 0.77, 9x9 1.68 -> 0.96, 15x15 ~20 -> 1.6, 23x23 49 -> 3.1 ms.
 * mv_gaussian_blur_u8_workspace_bytes() == 0:
 * the plain 2-D pass is as fast (fewer than 25 taps) or the size / width is outside the separable kernels (images narrower than 16
 * pixels ...) -- the call then IS mv_gaussian_blur_u8 and `workspace` may be NULL.  Taps must be non-negative with sum <= 1 (every Gaussian), else the 2-D
 * pass runs as well. */
int64_t mv_gaussian_blur_u8_workspace_bytes(int64_t planes, int h, int wdt, int kx, int ky);
int mv_gaussian_blur_u8_ws(const uint8_t* x, uint8_t* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx,
                           const float* k1d_y, int ky, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- separable filtering (BASELINE cfg3; two calls of the primitive fused into one kernel) ----
 * tmp = conv(pad_reflect(x), k1d_x as 1 x kx); y = conv(pad_reflect(tmp), k1d_y as ky x 1).
 * One HBM read + one HBM write per pixel; the intermediate lives in LDS. */
int mv_separable_blur_f32(const float* x, float* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx,
                          const float* k1d_y, int ky, void* stream);

/* fp16 / bf16 storage (the reference computes these dtypes natively, _misc.py:139-155): fp32 arithmetic in the 2-D tile
 * kernel, one round-to-nearest-even on store -- the bits of `.to(float32)` -> mv_gaussian_blur_f32's 2-D pass -> `.to(dtype)`
 * at 4 B per pixel of traffic instead of 20.  x, y: IEEE binary16 / bfloat16 planes.  Kernel sides up to 11
 * (MV_ERR_UNSUPPORTED beyond: convert and use the fp32 entry points). */
int mv_gaussian_blur_f16(const void* x, void* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx, const float* k1d_y,
                         int ky, void* stream);
int mv_gaussian_blur_bf16(const void* x, void* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx, const float* k1d_y,
                          int ky, void* stream);

/* float64 images.  The reference computes a float64 image in float64 -- taps (`_get_gaussian_kernel2d(..., dtype=dtype)`),
 * padding and conv2d (_misc.py:139-155); adjust_sharpness likewise (_color.py:246-275).  k1d_* are HOST arrays of doubles
 * (the reference's own float64 taps), kernel2d[j][i] = k1d_y[j] * k1d_x[i] in fp64; one fp64 fma chain per output in
 * row-major tap order.  mv_depthwise_conv2d_f64 takes the (ky, kx) taps as a DEVICE array (any odd size the 64 x 16 LDS
 * tile + halo holds: up to 63 x 63 and beyond for narrow kernels). */
int mv_gaussian_blur_f64(const double* x, double* y, int64_t planes, int h, int wdt, const double* k1d_x, int kx,
                         const double* k1d_y, int ky, void* stream);
int mv_depthwise_conv2d_f64(const double* x, double* y, const double* w_dev, int64_t planes, int h, int wdt, int ky, int kx,
                            int border, void* stream);
int mv_sharpness_f64(const double* x, double* y, int64_t planes, int h, int wdt, double sharpness_factor, int v1, void* stream);

/* uint8 storage, separable form (kernel sides up to 63, e.g. SimCLR-style GaussianBlur(23) on uint8 images): the separable
 * pair in fp32, then round_() and narrow (sides <= 7, and 9 x 9 / 9 x 7 / 7 x 9, on the 16-pixel-per-lane register kernel for
 * W >= 16; larger ones on the streaming kernel).  The reference evaluates one 2-D fp32 sum; the two differ by at most one
 * fp32 ulp before rounding, i.e. the uint8 results agree except at exact rounding ties (within the reference's own
 * atol = 1 for this op, test_transforms_v2.py:3309).  MV_ERR_UNSUPPORTED for sides <= 7 on images narrower than 16 pixels: use
 * mv_gaussian_blur_u8 there. */
int mv_separable_blur_u8(const uint8_t* x, uint8_t* y, int64_t planes, int h, int wdt, const float* k1d_x, int kx,
                         const float* k1d_y, int ky, void* stream);

/* ---- Sobel gradient (BASELINE cfg3; the primitive with taps [[-1,0,1],[-2,0,2],[-1,0,1]] and
 * its transpose, both outputs from one read of x). */
int mv_sobel_f32(const float* x, float* gx, float* gy, int64_t planes, int h, int wdt, int border, void* stream);
/* cfg3 graph fused: separable Gaussian (reflect) then Sobel (reflect) of the blurred image;
 * x is read once, gx and gy are written once (36 B / pixel). */
int mv_gaussian_sobel_f32(const float* x, float* gx, float* gy, int64_t planes, int h, int wdt, const float* k1d_x,
                          int kx, const float* k1d_y, int ky, void* stream);

/* ---- adjust_sharpness_image (_color.py:229-280; v1 _functional_tensor.py:809-838) -------------
 * Valid 3x3 smoothing [[1,1,1],[1,5,1],[1,1,1]]/13, integer inputs rounded, blended with the
 * input (v2: x + (1-f)*(blur-x) as one fma; v1: f*x + (1-f)*blur), clamped to [0, bound];
 * border pixels pass through.  `sharpness_factor` is the Python double; the library narrows
 * (1 - f) to float exactly as ATen does.  H <= 2 or W <= 2 copies the input.
 * The f32 entry also serves the reference's other integer dtypes after a caller-side .to(float32):
 * `bound` is _max_value(dtype) (1.0 for floating images) and `integer_semantics` != 0 rounds the
 * blurred value half-to-even before the blend, as the integer path does (_color.py:261-263). */
int mv_sharpness_f32(const float* x, float* y, int64_t planes, int h, int wdt, double sharpness_factor, int v1,
                     float bound, int integer_semantics, void* stream);
int mv_sharpness_u8(const uint8_t* x, uint8_t* y, int64_t planes, int h, int wdt, double sharpness_factor, int v1,
                    void* stream);

/* ---- separately allocated frames, ONE launch (the samples a DataLoader hands to Transform.forward,
 * transforms/v2/_transform.py:40-55, are allocated one by one; a launch per 1080p frame is launch-bound).
 * xs / ys: HOST arrays of `nframes` DEVICE pointers, frame i being `planes_per_frame` contiguous h x w planes; every frame has
 * the same shape.  The pointers travel by value in the kernel arguments (no device allocation, no copy, graph-capturable);
 * lists longer than 112 frames take one launch per 112.  Results are those of the single-frame entry points, bit for bit.
 * Frame pointers that are not 16-byte aligned are served by one launch per frame. */
int mv_gaussian_blur_f32_v(const float* const* xs, float* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt,
                           const float* k1d_x, int kx, const float* k1d_y, int ky, void* stream);
int mv_gaussian_blur_u8_v(const uint8_t* const* xs, uint8_t* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt,
                          const float* k1d_x, int kx, const float* k1d_y, int ky, void* stream);
int mv_separable_blur_f32_v(const float* const* xs, float* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt,
                            const float* k1d_x, int kx, const float* k1d_y, int ky, void* stream);
int mv_separable_blur_u8_v(const uint8_t* const* xs, uint8_t* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt,
                           const float* k1d_x, int kx, const float* k1d_y, int ky, void* stream);
int mv_sharpness_f32_v(const float* const* xs, float* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt,
                       double sharpness_factor, int v1, float bound, int integer_semantics, void* stream);
int mv_sharpness_u8_v(const uint8_t* const* xs, uint8_t* const* ys, int nframes, int64_t planes_per_frame, int h, int wdt,
                      double sharpness_factor, int v1, void* stream);

/* ---- first CNN layer: nn.Conv2d(cin, cout, 3, padding=1) [+ bias] [+ ReLU] (vgg.py:81-85,
 * ops/misc.py:97-119).  x (n,cin,h,w), w (cout,cin,3,3) and b (cout, may be NULL) are DEVICE
 * pointers (they are model parameters); y (n,cout,h,w).  Implicit GEMM on the fp32 MFMA
 * (v_mfma_f32_32x32x2_f32): exact fp32, K = cin*9 in (ci,dy,dx) order, bias as the last tap.  cin = 3 takes the
 * first-layer kernel (HBM-write-bound); any other cin the K-chunked kernel (MFMA-bound; feature maps up to 510
 * pixels wide). */
int mv_conv3x3_bias_relu_f32(const float* x, const float* w, const float* b, float* y, int64_t n, int cin, int h,
                             int wdt, int cout, int relu, void* stream);

/* ---- rest of the small CNNs' feature extractor ---------------------------------
 * nn.MaxPool2d(kernel_size=2, stride=2) (vgg.py:78-79): y is planes x (h/2) x (w/2), floor mode, NaN propagates. */
int mv_maxpool2x2_f32(const float* x, float* y, int64_t planes, int h, int wdt, void* stream);
/* nn.AdaptiveAvgPool2d((oh, ow)) (vgg.py:41): y is planes x oh x ow. */
int mv_adaptive_avgpool_f32(const float* x, float* y, int64_t planes, int h, int wdt, int oh, int ow, void* stream);

/* ---- Conv2dNormActivation (ops/misc.py:68-128) for the MobileNet family -----------------------
 * One kernel per block: conv2d (zero padding = (k-1)/2, no dilation) -> `+ bias` (may be NULL: the reference drops the
 * conv bias when a norm follows) -> folded norm -> `+ residual` (NULL or a tensor shaped like y: InvertedResidual's
 * `x + self.conv(x)`, models/mobilenetv2.py:61-63) -> activation.  All pointers are DEVICE pointers.
 *   kind    MV_CONV_DENSE3X3  w (cout, cin, 3, 3), cin <= 4: the stem (mobilenetv2.py:126)
 *           MV_CONV_DW3X3     w (c, 1, 3, 3), cin == cout == c: depthwise blocks (mobilenetv2.py:43-50)
 *           MV_CONV_PW1X1     w (cout, cin, 1, 1): pointwise convs (mobilenetv2.py:39-41, 52-53); fp32 MFMA
 *   stride  1 or 2 (1 for MV_CONV_PW1X1);  y is (n, cout, (h-1)/stride+1, (w-1)/stride+1)
 *   affine  MV_AFFINE_NONE;  MV_AFFINE_MUL_ADD = FrozenBatchNorm2d.forward (ops/misc.py:52-61): x*alpha then +beta
 *           (two roundings);  MV_AFFINE_FMA = nn.BatchNorm2d in eval mode (ATen): fma(x, alpha, beta) with
 *           (alpha, beta) from mv_fold_batchnorm
 *   act     MV_ACT_NONE / RELU / RELU6 / HARDSWISH / SILU */
#define MV_CONV_DENSE3X3 0
#define MV_CONV_DW3X3 1
#define MV_CONV_PW1X1 2
#define MV_AFFINE_NONE 0
#define MV_AFFINE_MUL_ADD 1
#define MV_AFFINE_FMA 2
#define MV_ACT_NONE 0
#define MV_ACT_RELU 1
#define MV_ACT_RELU6 2
#define MV_ACT_HARDSWISH 3
#define MV_ACT_SILU 4
int mv_conv_norm_act_f32(int kind, const float* x, const float* w, const float* bias, const float* alpha, const float* beta,
                         const float* residual, float* y, int64_t n, int cin, int h, int wdt, int cout, int stride,
                         int affine, int act, void* stream);
/* Summation order of a MV_CONV_PW1X1 call of this shape.  Large launches sum every output as ONE ascending-channel chain
 * (return value 1, *slice_len = cin).  Launches of few workgroups with long K (MobileNet's 7x7 / 14x14 layers) are latency-
 * bound as one chain; the kernel then walks K in `return value` slices of *slice_len channels (a multiple of 32; the last
 * slice may be shorter) with separate wave groups of one workgroup: each slice is an ascending chain from +0, the slices are
 * added in ascending order, then bias / norm / residual / activation.  Deterministic (no atomics, no workspace); within
 * 1e-6 relative of the single chain; oracle/oracle.c restates it (orc_pointwise_sliced_affine_act_f32). */
int mv_conv1x1_k_slices(int64_t n, int cin, int h, int wdt, int cout, int* slice_len);
/* ---- InvertedResidual (models/mobilenetv2.py:39-63) as ONE kernel -------------------------------------------------------
 * [1x1 expand cin -> hidden + norm + ReLU6] -> 3x3 depthwise (stride 1 | 2, zero padding 1) + norm + ReLU6 -> 1x1 project
 * hidden -> cout + norm [-> + x when `residual` (stride 1, cin == cout)].  The hidden tensor never reaches HBM: a workgroup
 * owns a region of output pixels and a slice of the hidden channels, expands a chunk of 32 channels on the fp32 MFMA into
 * LDS, runs the depthwise conv there and accumulates the projection in registers (csrc/invres.hip).
 *   x (n, cin, h, w), w_expand (hidden, cin), w_dw (hidden, 3, 3), w_project (cout, hidden), y (n, cout, oh, ow); a*, b*:
 *   the folded norms (`affine` = MV_AFFINE_MUL_ADD | MV_AFFINE_FMA for all three); all DEVICE pointers, 16-byte aligned.
 *   A block without expansion (the reference's expand_ratio == 1, mobilenetv2.py:37-38: hidden == cin, the depthwise conv runs
 *   on x itself) passes w_expand = a1 = b1 = NULL.
 * mv_inverted_residual_k_slices: 0 = no fused kernel for this shape (covered: square 28 / 14 / 7-pixel maps with MobileNetV2's
 * channel counts; 56-pixel maps with cin 24 and 112-pixel maps with cin 16 at stride 2 for any hidden % 8 == 0 and cout <= 32;
 * the 32 -> 32 -> cout <= 32 block without expansion on 112-pixel maps.  Else run the block as three mv_conv_norm_act_f32
 * calls); otherwise the number of slices of the projection's
 * summation order: expansion and depthwise conv are single ascending chains per output as in mv_conv_norm_act_f32; the
 * projection sums `return value` chains over *slice_len hidden channels each (ascending from +0; the last may be shorter),
 * adds them in ascending slice order, then norm, then `+ x`.  The plan depends on n (workgroup count), like
 * mv_conv3x3_k_slices.  More than one slice needs `workspace` of mv_inverted_residual_workspace_bytes() bytes (raw partial
 * sums, added by a second kernel of the same call); no atomics, deterministic.  oracle/oracle.c restates the block as
 * orc_conv2d_affine_act_f32 x 2 + orc_pointwise_sliced_affine_act_f32. */
int mv_inverted_residual_k_slices(int64_t n, int cin, int hidden, int cout, int h, int wdt, int stride, int* slice_len);
int64_t mv_inverted_residual_workspace_bytes(int64_t n, int cin, int hidden, int cout, int h, int wdt, int stride);
int mv_inverted_residual_f32(const float* x, const float* w_expand, const float* a1, const float* b1, const float* w_dw,
                             const float* a2, const float* b2, const float* w_project, const float* a3, const float* b3,
                             int residual, float* y, int64_t n, int cin, int hidden, int cout, int h, int wdt, int stride,
                             int affine, void* workspace, int64_t workspace_bytes, void* stream);
/* nn.BatchNorm2d(eval) -> (alpha, beta), HOST arrays in and out (weight / bias may be NULL = 1 / 0):
 * alpha = (1 / sqrt(var + eps)) * weight, beta = fma(-mean, alpha, bias) -- ATen batch_norm_cpu's own fp32 steps. */
void mv_fold_batchnorm(const float* weight, const float* bias, const float* mean, const float* var, double eps, int c,
                       float* alpha, float* beta);

/* ---- torchvision::deform_conv2d forward --------------------------------------------------------
 * The operator the reference registers with `TORCH_LIBRARY_FRAGMENT(torchvision, m)` (csrc/ops/deform_conv2d.cpp:164-169:
 * deform_conv2d(Tensor input, Tensor weight, Tensor offset, Tensor mask, Tensor bias, int stride_h, int stride_w, int pad_h,
 * int pad_w, int dilation_h, int dilation_w, int groups, int offset_groups, bool use_mask) -> Tensor); same arguments in
 * the same order, tensors as contiguous fp32 device pointers plus their sizes:
 *   x (n, cin, h, w), weight (cout, cin / groups, kh, kw), offset (n, 2 * offset_groups * kh * kw, oh, ow),
 *   mask (n, offset_groups * kh * kw, oh, ow) or NULL with use_mask = 0, bias (cout) or NULL, y (n, cout, oh, ow),
 *   oh = (h + 2*pad_h - (dilation_h*(kh-1) + 1)) / stride_h + 1, ow likewise.
 * One kernel (deformable gather fused into the GEMM's operand staging, no columns in memory) for every geometry whose tiles fit
 * in LDS -- kh*kw <= 40 and ordinary stride x dilation; then `workspace` may be NULL.  Otherwise `workspace` is device scratch
 * for the deformable im2col columns, at least mv_deform_conv2d_workspace_bytes(1, ...) (one image); the batch is processed in
 * passes of as many images as it holds.  mv_deform_conv2d_needs_workspace() says which: 0 = the fused kernel runs and no
 * workspace is used; 1 = the geometry needs one; 2 = optional: the launch is too small to fill the chip with the fused kernel's
 * tiles, the two-kernel form is faster if a workspace is passed (with NULL it runs fused).  Both forms compute the same chain
 * per output (bit-identical results). */
int64_t mv_deform_conv2d_workspace_bytes(int64_t images, int cin, int h, int wdt, int kh, int kw, int stride_h, int stride_w,
                                         int pad_h, int pad_w, int dilation_h, int dilation_w);
int mv_deform_conv2d_needs_workspace(int64_t images, int cin, int cout, int h, int wdt, int kh, int kw, int stride_h, int stride_w,
                                     int pad_h, int pad_w, int dilation_h, int dilation_w, int groups, int offset_groups);
int mv_deform_conv2d_f32(const float* x, const float* weight, const float* offset, const float* mask, const float* bias, float* y,
                         int64_t n, int cin, int h, int wdt, int cout, int kh, int kw, int stride_h, int stride_w, int pad_h,
                         int pad_w, int dilation_h, int dilation_w, int groups, int offset_groups, int use_mask, void* workspace,
                         int64_t workspace_bytes, void* stream);

/* ---- any other nn.Conv2d of the small CNNs (AlexNet's 11x11 stride 4 and 5x5, models/alexnet.py:22-33) ------------
 * conv2d(zero padding, stride, dilation, groups) + bias (may be NULL) + activation (MV_ACT_*) as an IMPLICIT GEMM on the fp32
 * MFMA: the pointwise kernel gathers each K chunk's im2col columns from the input while it stages them, so the columns exist
 * in LDS only -- never in HBM -- and `workspace` is not needed (NULL / 0).  One accumulator per output in ascending
 * (channel, ky, kx) order, then `+ bias`, like every conv of this library.  mv_conv2d_needs_workspace() returns 0 for every
 * geometry the implicit kernel covers (K = cin/groups * kh * kw < 65536, oh * ow < 2^20); 1 beyond that: a pointwise GEMM then
 * reads columns written by a plain im2col pass into `workspace` (at least mv_deform_conv2d_workspace_bytes(1, ...), one image;
 * more images per pass with more bytes); 2 = optional: the launch is a handful of workgroups (batch 1) and the columns form,
 * which cuts smaller tiles, is faster when the caller brings the workspace -- bit-identical results in every case. */
int mv_conv2d_needs_workspace(int64_t n, int cin, int cout, int h, int wdt, int kh, int kw, int stride_h, int stride_w, int pad_h,
                              int pad_w, int dilation_h, int dilation_w, int groups);
int mv_conv2d_bias_act_f32(const float* x, const float* weight, const float* bias, float* y, int64_t n, int cin, int h, int wdt,
                           int cout, int kh, int kw, int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h,
                           int dilation_w, int groups, int act, void* workspace, int64_t workspace_bytes, void* stream);
/* nn.MaxPool2d(kernel_size=k, stride=stride) without padding, floor mode (AlexNet's 3x3 stride 2): y is planes x
 * ((h-k)/stride+1) x ((w-k)/stride+1). */
int mv_maxpool2d_f32(const float* x, float* y, int64_t planes, int h, int wdt, int k, int stride, void* stream);

/* ---- the step BEFORE the path: ImageClassification's tail, transforms/_presets.py:58-60 -------
 * convert_image_dtype(float) = image.to(float32).mul_(1/255) (_misc.py:286-288), then normalize =
 * image.sub(mean).div_(std) (_misc.py:54-66).  x is (n, c, hw) planar; mean / std are HOST arrays of c floats
 * (c <= 16); NULL mean and std = conversion only. */
int mv_to_float_normalize_u8(const uint8_t* x, float* y, int64_t n, int c, int64_t hw, const float* mean, const float* stdv,
                             void* stream);
int mv_normalize_f32(const float* x, float* y, int64_t n, int c, int64_t hw, const float* mean, const float* stdv,
                     void* stream);
/* ---- the head of the same preset: F.resize(bilinear, antialias=True) + F.center_crop ---------------------------
 * (transforms/_presets.py:56-57; tensor path transforms/_functional_tensor.py:441-474 = .to(float32) ->
 * torch interpolate(bilinear, align_corners=False, antialias=True) -> torch.round + narrow for uint8;
 * transforms/functional.py:556-594 for the crop offsets and its zero padding).
 * x is (planes, h, w); the image is resized to (oh, ow) and y receives the (crop_h, crop_w) window whose top-left
 * corner is (crop_top, crop_left) in resized coordinates -- pass (0, 0, oh, ow) for a plain resize; parts of the
 * window outside the resized image are zero (center_crop's padding).  Only the rows and columns the window needs
 * are computed.  Scale factors up to 7 on both axes run as ONE kernel (width-pass rows in LDS): mv_resize_workspace_bytes(...)
 * returns 0 and `workspace` may be NULL.  Beyond, `workspace` is device scratch of at least that many bytes (fp32 width-pass
 * rows) and the call makes two launches on `stream`.  It allocates nothing; the results are the same bits either way. */
int64_t mv_resize_workspace_bytes(int64_t planes, int h, int wdt, int oh, int ow, int crop_top, int crop_left, int crop_h,
                                  int crop_w);
int mv_resize_bilinear_aa_u8(const uint8_t* x, uint8_t* y, int64_t planes, int h, int wdt, int oh, int ow, int crop_top,
                             int crop_left, int crop_h, int crop_w, void* workspace, int64_t workspace_bytes, void* stream);
int mv_resize_bilinear_aa_f32(const float* x, float* y, int64_t planes, int h, int wdt, int oh, int ow, int crop_top,
                              int crop_left, int crop_h, int crop_w, void* workspace, int64_t workspace_bytes, void* stream);
/* The whole ImageClassification.forward on a tensor image (transforms/_presets.py:56-63) in the same two launches:
 * resize -> center_crop -> convert_image_dtype(float) (v1: uint8 `.to(float32) / 255.0`,
 * _functional_tensor.py:93-99) -> normalize (`sub_(mean).div_(std)`, :928).  x is (n, c, h, w) with c <= 4,
 * y is fp32 (n, c, crop_h, crop_w); mean / std are HOST arrays of c floats. */
int mv_preset_classification_u8(const uint8_t* x, float* y, int64_t n, int c, int h, int wdt, int oh, int ow, int crop_top,
                                int crop_left, int crop_h, int crop_w, const float* mean, const float* stdv,
                                void* workspace, int64_t workspace_bytes, void* stream);
int mv_preset_classification_f32(const float* x, float* y, int64_t n, int c, int h, int wdt, int oh, int ow, int crop_top,
                                 int crop_left, int crop_h, int crop_w, const float* mean, const float* stdv,
                                 void* workspace, int64_t workspace_bytes, void* stream);
/* The same convolution with K SLICES ACROSS WORKGROUPS for launches too small to fill the chip with one chain per output (VGG's
 * 512 -> 512 layers at batch 1: ~50 workgroups walking 128 K chunks each).  mv_conv3x3_k_slices() states the summation order for a
 * shape -- 1: the single chain of mv_conv3x3_bias_relu_f32; otherwise that many chains over `slice_channels` input channels each
 * (the last may be shorter), every one from +0 in (channel, ky, kx) order, added in ascending slice order, then `+ bias`, then
 * ReLU: within 1e-6 relative of the single chain.  `workspace`: device scratch of mv_conv3x3_workspace_bytes() bytes (0 when the
 * shape is not sliced; then the call is mv_conv3x3_bias_relu_f32).
 * BATCH DEPENDENCE: the plan is a function of the workgroup count, hence of n -- the same image may differ in its last bits
 * between a batch-1 and a batch-8 call of the _ws entry (the same holds for mv_linear_k_slices and mv_conv1x1_k_slices).
 * mv_conv3x3_bias_relu_f32 / mv_linear_bias_relu_f32 (no workspace) keep ONE chain per output for every n: callers that need
 * batch-invariant bits use those (Python: functional.BATCH_INVARIANT_SUMMATION = True).  tests/golden/k_slice_plans.json
 * pins the plans of the VGG / AlexNet / MobileNetV2 shapes. */
int mv_conv3x3_k_slices(int64_t n, int cin, int h, int wdt, int cout, int* slice_channels);
int64_t mv_conv3x3_workspace_bytes(int64_t n, int cin, int h, int wdt, int cout);
int mv_conv3x3_bias_relu_ws_f32(const float* x, const float* w, const float* b, float* y, int64_t n, int cin, int h, int wdt, int cout,
                                int relu, void* workspace, int64_t workspace_bytes, void* stream);
/* The same conversion + normalisation fused into the first layer's load: x is the uint8 (n,3,h,w) image, the
 * fp32 normalised tensor never exists in HBM.  cout <= 64, w % 4 == 0. */
int mv_conv3x3_bias_relu_u8norm_f32(const uint8_t* x, const float* mean3, const float* std3, const float* w, const float* b,
                                    float* y, int64_t n, int h, int wdt, int cout, int relu, void* stream);

/* nn.Linear(k, m) [+ bias] [+ ReLU] (vgg.py:42-50): x (n, k), w (m, k) as nn.Linear stores it, b (m) or NULL,
 * y (n, m); all device pointers.  fp32 MFMA, one ascending-k chain per output (no split-K), bias as the last tap. */
int mv_linear_bias_relu_f32(const float* x, const float* w, const float* b, float* y, int64_t n, int k, int m, int relu,
                            void* stream);
/* The same layer for inference-size batches, where the single pass leaves most of the chip idle (25088 -> 4096 at batch 64
 * is 64 workgroups): K is cut into mv_linear_k_slices() contiguous slices of *slice_len k values (a multiple of 32); each
 * slice is an ascending-k chain from +0 whose partial sums go to workspace[slice][n][m]; a second kernel adds the partials
 * in ascending slice order, then the bias, then the ReLU.  Deterministic (no atomics); within 1e-6 relative of the single
 * chain.  mv_linear_workspace_bytes() == 0 (large batches): identical to mv_linear_bias_relu_f32, workspace may be NULL. */
int mv_linear_k_slices(int64_t n, int k, int m, int* slice_len);
int64_t mv_linear_workspace_bytes(int64_t n, int k, int m);
int mv_linear_bias_relu_ws_f32(const float* x, const float* w, const float* b, float* y, int64_t n, int k, int m, int relu,
                               void* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355VISION_H */
