import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from cpu_vision_amd import ops
from tools.perf_vgg import timeit
g = torch.Generator(device="cuda").manual_seed(0)
for n, c, m, hw in ((8, 256, 256, 64), (32, 64, 64, 112)):
    x = torch.rand((n, c, hw, hw), generator=g, device="cuda")
    w = torch.randn((m, c, 3, 3), generator=g, device="cuda") * 0.05
    b = torch.rand(m, generator=g, device="cuda")
    off = torch.randn((n, 18, hw, hw), generator=g, device="cuda") * 1.5
    mask = torch.rand((n, 9, hw, hw), generator=g, device="cuda")
    for ws in (1 << 30, 160 << 20, 80 << 20, 40 << 20):
        ops.MAX_WORKSPACE_BYTES = ws
        ms = timeit(lambda: ops.deform_conv2d(x, off, w, b, padding=(1, 1), mask=mask), 7)
        print(f"{n}x{c}x{hw} ws {ws >> 20:5d} MB: {ms:7.3f} ms {2.0 * n * m * hw * hw * c * 9 / ms / 1e9:6.1f} TF", flush=True)
