#!/usr/bin/env python3
"""A/B + ablation of the general-cin 3x3 conv (csrc/conv3x3_gen.hip) over whole-library variants
(cpu-vision_amd/lib/libmi355vision_<name>.so, built with MV_BUILD_VARIANT / MV_HIPCC_EXTRA), interleaved rounds in one process.

    MV_BUILD_VARIANT=noload MV_HIPCC_EXTRA=-DMV_GEN_ABLATE=1 python -c "import __graft_entry__ as g; g.build()"
    python tools/ab_conv.py base noload nostore neither nomfma
A name of the form KEY=VALUE[,KEY=VALUE] is the base library with those environment knobs set around its launches
(the launchers read MV_CONV_SHAPE / MV_CONV_SPEC / MV_CONV_COLFAST at every call):
    python tools/ab_conv.py base MV_CONV_COLFAST=1 MV_CONV_SHAPE=2,MV_CONV_SPEC=1
--batch=N keeps only the cases of that batch size.
"""
import os
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
# knob sweeps run on the -DMV_TUNING build of the same sources (the product library reads no environment variable)
os.environ.setdefault("MI355VISION_LIB", str(Path(__file__).resolve().parent.parent / "cpu-vision_amd" / "lib" / "libmi355vision_tuning.so"))
import torch  # noqa: E402

names = [a for a in sys.argv[1:] if not a.startswith("--")] or ["base"]
only_batch = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--batch=")), None)
libs = {}
envs = {}
for n in names:
    if "=" in n:
        envs[n] = dict(kv.split("=", 1) for kv in n.split(","))
    # KEY=VALUE names need the -DMV_TUNING build: the product library reads no environment variable
    p = ROOT / "cpu-vision_amd" / "lib" / ("libmi355vision.so" if n == "base" else "libmi355vision_tuning.so" if "=" in n else f"libmi355vision_{n}.so")
    lib = C.CDLL(str(p))
    lib.mv_conv3x3_bias_relu_f32.argtypes = [C.c_void_p] * 4 + [C.c_int64] + [C.c_int] * 5 + [C.c_void_p]
    libs[n] = lib
g = torch.Generator(device="cuda").manual_seed(0)
s = torch.cuda.current_stream().cuda_stream
cases = [(1, 512, 512, 28), (1, 512, 512, 14), (1, 256, 256, 56), (8, 512, 512, 28), (8, 512, 512, 14), (8, 128, 256, 56), (64, 512, 512, 14),
         (64, 256, 256, 56), (64, 64, 128, 112), (64, 512, 512, 28), (64, 128, 256, 56), (64, 256, 512, 28)]
if only_batch is not None:
    cases = [c for c in cases if c[0] == only_batch]


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for n_img, cin, cout, hw in cases:
    x = torch.rand((n_img, cin, hw, hw), generator=g, device="cuda")
    w = torch.randn((cout, cin, 3, 3), generator=g, device="cuda") * 0.02
    b = torch.rand(cout, generator=g, device="cuda")
    y = torch.empty((n_img, cout, hw, hw), device="cuda")
    res = {n: [] for n in names}
    ref = None
    for r in range(10):
        for n, lib in libs.items():
            for k, v in envs.get(n, {}).items():
                os.environ[k] = v
            t = timed(lambda: lib.mv_conv3x3_bias_relu_f32(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), n_img, cin, hw, hw, cout, 1, s))
            for k in envs.get(n, {}):
                os.environ.pop(k, None)
            if r == 0 and n not in ("noload", "nostore", "neither", "neither0", "nomfma", "nomfma6"):  # every non-ablation variant computes the same bits
                if ref is None:
                    ref = y.clone()
                else:
                    assert torch.equal(ref, y), f"{n} changes the result"
            if r >= 2:
                res[n].append(t)
    line = f"conv {cin}->{cout} @{hw} batch {n_img:3d}:"
    for n in names:
        v = sorted(res[n])
        line += f"  {n} {v[len(v) // 2] * 1e3:7.1f} us"
    print(line, flush=True)
