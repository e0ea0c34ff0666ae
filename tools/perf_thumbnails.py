import sys
sys.path.insert(0, "/root/repo")
import torch
from cpu_vision_amd import functional as F
from tools.perf_configs import timeit
g = torch.Generator(device="cuda").manual_seed(0)
for shape in ((4096, 3, 32, 32), (1024, 3, 64, 64), (512, 3, 96, 96), (256, 3, 128, 128)):
    xf = torch.rand(shape, generator=g, device="cuda")
    xu = (xf * 255).to(torch.uint8)
    for name, x, bpp in (("u8 ", xu, 2), ("f32", xf, 8)):
        line = f"{shape[0]}x3x{shape[2]}x{shape[3]} {name}:"
        for op, fn in (("blur3", lambda: F.gaussian_blur(x, [3, 3])), ("blur5", lambda: F.gaussian_blur(x, [5, 5])), ("sharp", lambda: F.adjust_sharpness(x, 1.7))):
            ms, _ = timeit(fn, 7)
            line += f"  {op} {ms * 1e3:6.1f} us ({x.numel() * bpp / ms / 1e6:5.0f} GB/s)"
        print(line, flush=True)
