"""Copies the artifacts of scripts/gpu_full.sh and scripts/gpu_prof_scl.sh from gpurun_out/ into profiles/<tag>/
and recomputes profiles/traffic.json (HBM bytes per frame of the dominant SC kernel from the FETCH_SIZE /
WRITE_SIZE passes).  Usage: python tools/refresh_profiles.py [tag]   (default r01)
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def latest(pattern):
    files = glob.glob(os.path.join(OUT, pattern))
    if not files:
        raise SystemExit("missing " + pattern)
    return max(files, key=os.path.getmtime)


def per_launch(path, counter, kernel):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    if not vals:
        raise SystemExit("no %s rows for %s in %s" % (counter, kernel, path))
    return sum(vals) / len(vals), len(vals)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    bench = json.loads(open(os.path.join(OUT, "bench.json")).read().strip().splitlines()[-1])
    kernel = bench["roofline"]["kernel"]
    frames = bench["config"]["frames_per_gpu"]
    short = kernel.split("<")[0]
    shutil.copy(os.path.join(OUT, "bench.json"), os.path.join(dst, "bench_n1.json"))
    shutil.copy(os.path.join(OUT, "tests_gpu.log"), os.path.join(dst, "tests_gpu.log"))
    shutil.copy(latest("prof_stats/*/*kernel_stats.csv"), os.path.join(dst, "bench_kernel_stats.csv"))
    fpath, wpath = latest("pmcfull_FETCH_SIZE/*/*counter_collection.csv"), latest("pmcfull_WRITE_SIZE/*/*counter_collection.csv")
    shutil.copy(fpath, os.path.join(dst, "pmc_fetch_size.csv"))
    shutil.copy(wpath, os.path.join(dst, "pmc_write_size.csv"))
    fetch, nf = per_launch(fpath, "FETCH_SIZE", short)
    write, nw = per_launch(wpath, "WRITE_SIZE", short)
    per_frame = (2 * fetch + write) * 1024 / frames
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    tab = json.load(open(tpath)) if os.path.exists(tpath) else {}
    tab[kernel] = per_frame
    tab["_source"] = {
        "kernel": kernel,
        "FETCH_SIZE_KiB_per_launch": fetch,
        "WRITE_SIZE_KiB_per_launch": write,
        "frames_per_launch": frames,
        "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 / frames (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)",
    }
    json.dump(tab, open(tpath, "w"), indent=1)
    try:
        shutil.copy(latest("prof_scl/*/*kernel_stats.csv"), os.path.join(dst, "scl8_kernel_stats.csv"))
    except SystemExit:
        pass
    print("kernel %s: FETCH %.0f KiB (%d launches), WRITE %.0f KiB (%d launches) -> %.0f B per frame" % (kernel, fetch, nf, write, nw, per_frame))
    print("bench: %.1f ms per step, %.0f M frames/s, kernel %.2f ms" % (bench["ms_per_step"], bench["frames_per_s"] / 1e6, bench["roofline"]["kernel_ms"]))
    for r in csv.DictReader(open(os.path.join(dst, "bench_kernel_stats.csv"))):
        if short in r["Name"]:
            print("rocprof: %s calls, average %.3f ms" % (r["Calls"], float(r["AverageNs"]) / 1e6))


if __name__ == "__main__":
    main()
