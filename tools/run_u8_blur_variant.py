import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pathlib import Path
import torch
from cpu_vision_amd import functional as F, _lib
root = Path(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
x = torch.randint(0, 256, (32, 3, 2160, 3840), dtype=torch.uint8, device="cuda")
with _lib.tuning_library(root / "cpu-vision_amd" / "lib" / "libmi355vision_tfab.so"):
    for k, s in ((9, 1.7), (15, 2.6), (23, 3.8)):
        for _ in range(4):
            y = F.gaussian_blur(x, [k, k], [s, s])
torch.cuda.synchronize()
