"""mi355vision -- the MI355X (gfx950) implementation of CPU-Vision's image-filtering hot path.

Python host layer (the reference is Python) over the C ABI in include/mi355vision.h:
  functional      gaussian_blur / adjust_sharpness (+ _image/_video kernels), box / separable / Sobel, conv+ReLU
  functional_v1   the v1 tensor backend's gaussian_blur / adjust_sharpness, resize / center_crop
  mobilenet       Conv2dNormActivation with a folded norm, InvertedResidual, MobileNetV2
  ops             deform_conv2d / DeformConv2d (+ registration behind torch.ops.torchvision.deform_conv2d)
  graphs          HIP-graph capture of a whole forward (batch-1 latency)
  presets         ImageClassification (resize -> center_crop -> float -> normalize in one call)
  transforms      GaussianBlur, RandomAdjustSharpness, GaussianBlurV1
  nn              Conv3x3ReLU, VGG / AlexNet (re-exports mobilenet.Conv2dNormActivation: one class under that name)
  sharding        frame-block partitioning over the GPUs of a node (no data-path collective)
  register_kernel the reference's kernel-registry plugin surface

The directory is named `cpu-vision_amd`; import it as `cpu_vision_amd` (alias package at the repo root).
"""
from . import functional, functional_v1, graphs, mobilenet, nn, ops, presets, sharding, transforms, tv_tensors  # noqa: F401
from ._lib import LIB_PATH, Mi355VisionError, load as load_library  # noqa: F401
from ._registry import register_kernel  # noqa: F401
from .functional import (adjust_sharpness, adjust_sharpness_image, box_filter, conv2d_bias_relu,  # noqa: F401
                         depthwise_conv2d, gaussian_blur, gaussian_blur_image, gaussian_sobel,
                         separable_gaussian_blur, sobel)

__version__ = "0.1.0"
