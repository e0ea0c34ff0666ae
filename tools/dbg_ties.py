import sys, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from cpu_vision_amd import _lib, functional as F
lib = _lib.load()
g = torch.Generator(device="cuda").manual_seed(0)
for planes, h, w in ((3, 2160, 3840), (96, 2160, 3840)):
    x = torch.randint(0, 256, (planes, h, w), generator=g, device="cuda", dtype=torch.uint8)
    y = torch.empty_like(x)
    for k, sg in ((7, 1.4), (9, 1.7), (15, 2.6), (23, 3.8)):
        t = _lib.taps_from_tensor(F._get_gaussian_kernel1d(k, sg))
        nb = int(lib.mv_gaussian_blur_u8_workspace_bytes(planes, h, w, k, k))
        if nb == 0:  # up to 49 taps the plain 2-D pass runs (gaussian_blur_u8_hybrid_supported)
            continue
        ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")
        def run():
            _lib.check(lib.mv_gaussian_blur_u8_ws(x.data_ptr(), y.data_ptr(), planes, h, w, t, k, t, k, ws.data_ptr(), nb, None))
        run(); torch.cuda.synchronize()
        hdr = ws[:16].view(torch.int32).cpu().tolist()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        lane_rows = planes * h * ((w + hdr[2] - 1) // hdr[2])
        print(f"{planes}x{h}x{w} {k}x{k}: count {hdr[0]} capacity {hdr[1]} per segment x 64 npx {hdr[2]}  flagged {hdr[0] / lane_rows * 100:.2f} % of lane-rows  {e0.elapsed_time(e1):.3f} ms  {_lib.last_kernel()}")
