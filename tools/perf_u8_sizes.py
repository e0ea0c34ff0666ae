import sys
sys.path.insert(0, "/root/repo")
import torch
from cpu_vision_amd import functional as F
from tools.perf_configs import timeit
g = torch.Generator(device="cuda").manual_seed(0)
xu = torch.randint(0, 256, (32, 3, 2160, 3840), generator=g, device="cuda", dtype=torch.uint8)
for ks in ([9, 5], [1, 9], [9, 3], [3, 9], [5, 9], [11, 3], [7, 5], [1, 5], [1, 3]):
    ms, _ = timeit(lambda: F.gaussian_blur(xu, ks), 5)
    print(f"u8 gaussian_blur {ks[0]}x{ks[1]}: {ms:7.3f} ms", flush=True)
