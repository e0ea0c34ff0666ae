#!/usr/bin/env python3
"""HBM traffic of the headline kernel from PMC counters (runs on the GPU box; spawns rocprofv3 as children).

Two separate passes, as MI355X_MICROARCH.md prescribes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2: they do not
fit one pass), each `rocprofv3 --pmc <counter> --kernel-trace -- python3 bench.py ...`.  Corrections applied:
  * FETCH_SIZE / WRITE_SIZE are in KiB -> x1024;
  * on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane)
    -> x2 on the read side; WRITE_SIZE is exact for 16-B-per-lane streaming stores.
Writes profiles/traffic_latest.json (read by bench.py) and profiles/<tag>_pmc_traffic.txt.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
BENCH_LINE = {}
ALL_GROUPS = {}


def one_pass(counter, frames, outdir):
    outdir.mkdir(parents=True, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", str(outdir), "--",
           "python3", str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--frames-per-gpu", str(frames),
           "--config-launches", "4"]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-2000:])
        raise SystemExit(f"rocprofv3 pass {counter} failed")
    global BENCH_LINE
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            BENCH_LINE = json.loads(ln)
    # bench.py also runs the cfg2 / cfg3 / cfg4 legs: group the dispatches by (kernel, grid size) -- the headline is the k_dwtile
    # dispatch with the 128-frame grid, cfg2 the same kernel on a smaller one
    groups = {}
    for f in glob.glob(str(outdir / "**" / "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if "mv::k_" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                key = (row["Kernel_Name"].replace("void mv::", "").split("(")[0], int(row.get("Grid_Size", 0) or 0))
                groups.setdefault(key, []).append(float(row["Counter_Value"]))
    if not groups:
        raise SystemExit(f"no {counter} rows for the mv:: kernels")
    ALL_GROUPS[counter] = groups
    head = max((k for k in groups if k[0].startswith("k_dwtile")), key=lambda k: k[1], default=None)
    if head is None:
        raise SystemExit("no k_dwtile dispatch in the trace")
    return groups[head]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames-per-gpu", type=int, default=128)
    ap.add_argument("--tag", default="r01")
    a = ap.parse_args()
    scratch = ROOT / "gpurun_out" / "pmc"
    fetch = one_pass("FETCH_SIZE", a.frames_per_gpu, scratch / "fetch")
    write = one_pass("WRITE_SIZE", a.frames_per_gpu, scratch / "write")
    fetch_kib = sum(fetch) / len(fetch)
    write_kib = sum(write) / len(write)
    read_bytes = fetch_kib * 1024 * 2  # gfx950: FETCH_SIZE counts 128-B requests at 64 B
    write_bytes = write_kib * 1024
    alg = a.frames_per_gpu * 3 * 2160 * 3840 * 8
    out = {
        "frames_per_gpu": a.frames_per_gpu,
        "hbm_bytes_per_launch": int(read_bytes + write_bytes),
        "read_bytes_per_launch": int(read_bytes),
        "write_bytes_per_launch": int(write_bytes),
        "algorithmic_bytes_per_launch": alg,
        "traffic_over_algorithmic": round((read_bytes + write_bytes) / alg, 4),
        "raw": {"FETCH_SIZE_KiB_avg": fetch_kib, "WRITE_SIZE_KiB_avg": write_kib, "dispatches": len(fetch)},
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 gfx950 correction",
        # what the numbers were measured on: bench.py reports them only for the same kernel sources
        "kernel": BENCH_LINE.get("roofline", {}).get("kernel"),
        "library_build_id": BENCH_LINE.get("roofline", {}).get("library_build_id"),
        "kernel_source_sha": __import__("bench").kernel_source_sha(),
    }
    (ROOT / "profiles").mkdir(exist_ok=True)
    (ROOT / "profiles" / "traffic_latest.json").write_text(json.dumps(out, indent=1))
    (ROOT / "profiles" / f"{a.tag}_pmc_traffic.txt").write_text(
        f"kernel {out['kernel']} of library build {out['library_build_id']} (bench.py headline), {a.frames_per_gpu} frames of 3x2160x3840 fp32 per launch\n"
        f"FETCH_SIZE avg {fetch_kib:.1f} KiB  -> x1024 x2 = {read_bytes / 1e9:.3f} GB read\n"
        f"WRITE_SIZE avg {write_kib:.1f} KiB  -> x1024    = {write_bytes / 1e9:.3f} GB written\n"
        f"algorithmic {alg / 1e9:.3f} GB ; measured / algorithmic = {(read_bytes + write_bytes) / alg:.4f}\n")
    # gpurun only brings gpurun_out/ back from the GPU box: leave copies there (copy them into profiles/ and commit)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "traffic_latest.json").write_text(json.dumps(out, indent=1))
    (ROOT / "gpurun_out" / f"{a.tag}_pmc_traffic.txt").write_text((ROOT / "profiles" / f"{a.tag}_pmc_traffic.txt").read_text())
    # the other legs of the same two passes: per (kernel, grid) FETCH x 2 / WRITE against the leg's algorithmic bytes
    legs = BENCH_LINE.get("configs", {})
    lines = []
    for cfg, leg in legs.items():
        kname = leg["kernel"].split("<")[0]
        cands = [k for k in ALL_GROUPS.get("FETCH_SIZE", {}) if k[0].startswith(kname) and k in ALL_GROUPS.get("WRITE_SIZE", {})]
        if cfg == "cfg2":  # same kernel as the headline: the smaller grid
            cands = sorted(cands, key=lambda k: k[1])[:1]
        elif cands:
            cands = [max(cands, key=lambda k: k[1])]
        for k in cands:
            fe = ALL_GROUPS["FETCH_SIZE"][k]
            wr = ALL_GROUPS["WRITE_SIZE"][k]
            rd_b, wr_b = sum(fe) / len(fe) * 1024 * 2, sum(wr) / len(wr) * 1024
            alg_b = leg["algorithmic_bytes_per_launch"]
            lines.append(f"{cfg}: {k[0]} grid {k[1]}: FETCH_SIZE x2 = {rd_b / 1e9:.3f} GB read, WRITE_SIZE = {wr_b / 1e9:.3f} GB written, "
                         f"algorithmic {alg_b / 1e9:.3f} GB -> measured / algorithmic = {(rd_b + wr_b) / alg_b:.4f}  ({len(fe)} dispatches)")
    if lines:
        text = "\n".join(lines) + "\n"
        (ROOT / "profiles" / f"{a.tag}_pmc_traffic_configs.txt").write_text(text)
        (ROOT / "gpurun_out" / f"{a.tag}_pmc_traffic_configs.txt").write_text(text)
        print(text)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
