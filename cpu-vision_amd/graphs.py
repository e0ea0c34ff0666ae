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


class CapturedParallel:
    """Independent calls (one frame each, say) captured into ONE graph whose kernel nodes sit on `streams` parallel branches.

    A stream of one-frame launches pays ~3.5 us of ramp-up and tail per kernel (tools/perf_single_frame.py); on parallel
    branches consecutive frames overlap them and replaying costs no host work per frame: a 1080p frame per call
    13.8 -> 11.6 us with two branches (profiles/r01_perf_frames_graph.log).  One batched launch over contiguous frames stays
    the fastest form (8.4 us per frame) -- this is for frames that arrive as separate tensors.  Each call must launch on
    the current stream (every function of this package does) and touch only its own tensors."""

    def __init__(self, calls, streams: int = 2, warmup: int = 1, device=None) -> None:
        calls = list(calls)
        if not calls:
            raise ValueError("capture_parallel needs at least one call")
        if streams < 1:
            raise ValueError("streams should be >= 1")
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._side = [torch.cuda.Stream(device=dev) for _ in range(min(streams, len(calls)))]

        def issue():
            cur = torch.cuda.current_stream(dev)
            outs = []
            for s in self._side:
                s.wait_stream(cur)
            for i, call in enumerate(calls):
                with torch.cuda.stream(self._side[i % len(self._side)]):
                    outs.append(call())
            for s in self._side:
                cur.wait_stream(s)
            return outs

        for _ in range(warmup):  # loads the library, warms the allocator -- none of it may be captured
            issue()
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = issue()

    def replay(self):
        """Run every captured call again (on whatever the captured input tensors hold now); returns the static outputs."""
        self.graph.replay()
        return self.outputs


def capture_parallel(calls, streams: int = 2, warmup: int = 1, device=None) -> CapturedParallel:
    return CapturedParallel(calls, streams, warmup, device)
