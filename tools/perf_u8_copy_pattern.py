import ctypes as C, os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
fp = C.POINTER(C.c_float)
libs = {}
for n in ("abl2", "abl2plain"):
    lib = C.CDLL(f"{ROOT}/cpu-vision_amd/lib/libmi355vision_{n}.so")
    lib.mv_separable_blur_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, fp, C.c_int, fp, C.c_int, C.c_void_p]
    libs[n] = lib
k5 = (C.c_float * 5)(0.1, 0.2, 0.4, 0.2, 0.1)
s = torch.cuda.current_stream().cuda_stream
def timed(fn, n=12):
    ts = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); rc = fn(); e1.record(); torch.cuda.synchronize()
        assert rc == 0
        if i >= 2: ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]
for (planes, h, w) in ((96, 2160, 3840), (90, 2160, 4096), (96, 2160, 3072), (192, 1080, 3840)):
    x = torch.randint(0, 256, (planes, h, w), device="cuda", dtype=torch.uint8)
    y = torch.empty_like(x)
    nbytes = x.numel() * 2
    line = f"{planes} x {h} x {w}:"
    for n, lib in libs.items():
        for rows in ("", "32", "128"):
            if rows: os.environ["MV_DWK_U8_ROWS"] = rows
            else: os.environ.pop("MV_DWK_U8_ROWS", None)
            ms = timed(lambda: lib.mv_separable_blur_u8(x.data_ptr(), y.data_ptr(), planes, h, w, k5, 5, k5, 5, s))
            line += f"  {n}/rows{rows or 64} {nbytes / ms / 1e6:5.0f} GB/s"
    print(line, flush=True)
    yy = torch.empty_like(x)
    ms = timed(lambda: (yy.copy_(x), 0)[1])
    print(f"    torch copy_ of the same uint8 tensor: {nbytes / ms / 1e6:5.0f} GB/s", flush=True)
    del x, y, yy
