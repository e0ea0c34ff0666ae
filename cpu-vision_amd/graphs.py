"""HIP-graph capture of a whole forward pass (launch-bound small batches).

Every entry point of libmi355vision.so is asynchronous on the stream it is given, allocates nothing and never
synchronises, so a forward built from them can be captured once into a HIP graph (torch.cuda.CUDAGraph on ROCm) and
replayed: one graph launch instead of ~55 kernel launches + their Python (MobileNetV2), which is what bounds the
latency of a batch-1 request.  The Python wrappers' output / workspace allocations happen at capture time in the
graph's private pool and are reused by every replay.
"""
from __future__ import annotations

from typing import Callable

import torch


class CapturedForward:
    """fn(static_input) captured once; `__call__(x)` copies x into the static input, replays, returns the static output
    (valid until the next call; clone it to keep it)."""

    def __init__(self, fn: Callable[[torch.Tensor], torch.Tensor], example: torch.Tensor, warmup: int = 2) -> None:
        if not example.is_cuda:
            raise ValueError("capture needs a device tensor")
        self.static_in = example.clone()
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):  # folds norms, loads the library, warms the allocator -- none of it may be captured
                fn(self.static_in)
        torch.cuda.current_stream(example.device).wait_stream(side)
        torch.cuda.synchronize(example.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = fn(self.static_in)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype:
            raise ValueError(f"captured for {tuple(self.static_in.shape)} {self.static_in.dtype}, got {tuple(x.shape)} {x.dtype}")
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out


def capture(fn: Callable[[torch.Tensor], torch.Tensor], example: torch.Tensor, warmup: int = 2) -> CapturedForward:
    return CapturedForward(fn, example, warmup)
