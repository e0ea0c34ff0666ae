// deform_conv2d_shim.cpp -- boundary B2 built for real: the first NON-Python consumer of the C ABI.
//
// The reference attaches device back ends to its operators through PyTorch's C++ dispatcher:
//   schema    torchvision::deform_conv2d(...)                      csrc/ops/deform_conv2d.cpp:164-169
//   CUDA key  TORCH_LIBRARY_IMPL(torchvision, CUDA, m)             csrc/ops/cuda/deform_conv2d_kernel.cu:1323
//   autocast  inputs to float32, result back to the input's dtype  csrc/ops/autocast/deform_conv2d_kernel.cpp:12-52
//   fake      output shape for tracing (Python there)              torchvision/_meta_registrations.py:177-198
// This file is what a maintainer adds next to csrc/ops/cuda/: the same three registrations over libmi355vision.so.  It sees
// the library only through include/mi355vision.h (plain pointers and sizes) and PyTorch only through its public C++ API; it
// is compiled by cpu-vision_amd/torch_shim/build.py (torch.utils.cpp_extension, g++) and linked against libmi355vision.so.
// On a ROCm build of PyTorch the "CUDA" dispatch key IS the HIP device.
#include <ATen/ATen.h>
#include <ATen/autocast_mode.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <algorithm>

#include "mi355vision.h"

namespace {

struct Geometry {
  int64_t n, cin, h, w, cout, kh, kw, oh, ow;
};

Geometry geometry_of(const at::Tensor& input, const at::Tensor& weight, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw) {
  Geometry g{input.size(0), input.size(1), input.size(2), input.size(3), weight.size(0), weight.size(2), weight.size(3), 0, 0};
  g.oh = (g.h + 2 * ph - (dh * (g.kh - 1) + 1)) / sh + 1;
  g.ow = (g.w + 2 * pw - (dw * (g.kw - 1) + 1)) / sw + 1;
  return g;
}

// The device kernel behind the dispatcher.  float32 is the arithmetic of the library; float64 / float16 / bfloat16 tensors (the
// reference's own tests run the op in float64, test/test_ops.py:931) are computed in float32 and returned in their dtype,
// like the Python layer of this package does.
at::Tensor deform_conv2d_mi355(const at::Tensor& input, const at::Tensor& weight, const at::Tensor& offset, const at::Tensor& mask,
                               const at::Tensor& bias, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw,
                               int64_t groups, int64_t offset_groups, bool use_mask) {
  TORCH_CHECK(input.dim() == 4 && weight.dim() == 4 && offset.dim() == 4, "deform_conv2d: input, weight and offset must be 4-D");
  TORCH_CHECK(!use_mask || mask.dim() == 4, "deform_conv2d: mask must be 4-D");
  TORCH_CHECK(weight.is_cuda() && offset.is_cuda(), "deform_conv2d: every tensor must live on the MI355X (there is no CPU path here)");
  const c10::hip::HIPGuard device_guard(input.device().index());
  const auto f32 = [](const at::Tensor& t) { return t.to(at::kFloat).contiguous(); };
  const at::Tensor x = f32(input), w = f32(weight), off = f32(offset);
  const at::Tensor m = use_mask ? f32(mask) : at::Tensor(), b = bias.numel() ? f32(bias) : at::Tensor();
  const Geometry g = geometry_of(x, w, sh, sw, ph, pw, dh, dw);
  TORCH_CHECK(g.oh > 0 && g.ow > 0, "deform_conv2d: calculated output size too small - out_h: ", g.oh, " out_w: ", g.ow);
  at::Tensor y = at::empty({g.n, g.cout, g.oh, g.ow}, x.options());
  if (y.numel() == 0) return y.to(input.scalar_type());
  // 0: the fused kernel needs no scratch; 1: this geometry runs through a columns workspace; 2: optional (faster for tiny launches)
  at::Tensor scratch;
  if (mv_deform_conv2d_needs_workspace(g.n, (int)g.cin, (int)g.cout, (int)g.h, (int)g.w, (int)g.kh, (int)g.kw, (int)sh, (int)sw, (int)ph,
                                       (int)pw, (int)dh, (int)dw, (int)groups, (int)offset_groups) != 0) {
    const int64_t bytes = mv_deform_conv2d_workspace_bytes(std::min<int64_t>(g.n, 32), (int)g.cin, (int)g.h, (int)g.w, (int)g.kh, (int)g.kw,
                                                           (int)sh, (int)sw, (int)ph, (int)pw, (int)dh, (int)dw);
    scratch = at::empty({bytes}, x.options().dtype(at::kByte));
  }
  const int rc = mv_deform_conv2d_f32(x.data_ptr<float>(), w.data_ptr<float>(), off.data_ptr<float>(), use_mask ? m.data_ptr<float>() : nullptr,
                                      b.defined() ? b.data_ptr<float>() : nullptr, y.data_ptr<float>(), g.n, (int)g.cin, (int)g.h, (int)g.w,
                                      (int)g.cout, (int)g.kh, (int)g.kw, (int)sh, (int)sw, (int)ph, (int)pw, (int)dh, (int)dw, (int)groups,
                                      (int)offset_groups, use_mask ? 1 : 0, scratch.defined() ? scratch.data_ptr() : nullptr,
                                      scratch.defined() ? scratch.numel() : 0, c10::hip::getCurrentHIPStream().stream());
  TORCH_CHECK(rc == 0, "libmi355vision: ", mv_last_error());
  return y.to(input.scalar_type());
}

// Shape rule for tracing (FakeTensor / torch.compile / torch.export): (N, C_out, offset.H, offset.W), symbolic sizes kept.
at::Tensor deform_conv2d_meta(const at::Tensor& input, const at::Tensor& weight, const at::Tensor& offset, const at::Tensor&, const at::Tensor&,
                              c10::SymInt, c10::SymInt, c10::SymInt, c10::SymInt, c10::SymInt, c10::SymInt, c10::SymInt, c10::SymInt, bool) {
  return input.new_empty_symint({input.sym_size(0), weight.sym_size(0), offset.sym_size(2), offset.sym_size(3)});
}

// Under torch.autocast the operator keeps float32 arithmetic: every floating tensor argument is widened through autocast's cast
// cache, the op is re-dispatched below the autocast key, and the result takes the input's dtype again.
at::Tensor deform_conv2d_autocast(const at::Tensor& input, const at::Tensor& weight, const at::Tensor& offset, const at::Tensor& mask,
                                  const at::Tensor& bias, c10::SymInt sh, c10::SymInt sw, c10::SymInt ph, c10::SymInt pw, c10::SymInt dh,
                                  c10::SymInt dw, c10::SymInt groups, c10::SymInt offset_groups, bool use_mask) {
  const c10::impl::ExcludeDispatchKeyGuard below_autocast(c10::DispatchKey::Autocast);
  const auto wide = [](const at::Tensor& t) { return at::autocast::cached_cast(at::kFloat, t); };
  static const auto op = c10::Dispatcher::singleton().findSchemaOrThrow("torchvision::deform_conv2d", "")
                             .typed<at::Tensor(const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&, const at::Tensor&,
                                               c10::SymInt, c10::SymInt, c10::SymInt, c10::SymInt, c10::SymInt, c10::SymInt, c10::SymInt,
                                               c10::SymInt, bool)>();
  return op.call(wide(input), wide(weight), wide(offset), wide(mask), wide(bias), sh, sw, ph, pw, dh, dw, groups, offset_groups, use_mask)
      .to(input.scalar_type());
}

}  // namespace

TORCH_LIBRARY_IMPL(torchvision, CUDA, m) { m.impl("deform_conv2d", TORCH_FN(deform_conv2d_mi355)); }
TORCH_LIBRARY_IMPL(torchvision, Meta, m) { m.impl("deform_conv2d", TORCH_FN(deform_conv2d_meta)); }
TORCH_LIBRARY_IMPL(torchvision, Autocast, m) { m.impl("deform_conv2d", TORCH_FN(deform_conv2d_autocast)); }
