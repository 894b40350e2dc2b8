"""Collects the artefacts of one measurement pass from gpurun_out/ into profiles/<tag>/ and rewrites profiles/traffic.json,
all from one commit.  The pass is made of scripts/gpu_full.sh (GPU tests, smoke, bench.py, rocprofv3 --kernel-trace --stats of
the bench, FETCH_SIZE / WRITE_SIZE passes of the bench), scripts/gpu_prof_legs.sh (kernel times, SQ counters and traffic of
the secondary kernels), scripts/gpu_pmc3.sh (issue / wait counters of the bench's SC kernel), the two-rank rehearsal of
bench.py and the cycle profile of the register list decoder; run them on the committed tree, then:

    python tools/refresh_profiles.py r03

profiles/traffic.json maps the kernel names polar_last_launch reports to {bytes_per_frame, what, frames_per_launch, file,
commit, date}: bench.py looks its `traffic` figures up there and prints the entry's origin as `traffic_source`.
"""
import csv
import datetime
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")

# kernel names as rocprofv3 prints them -> as polar_last_launch reports them (the key bench.py looks up)
LEG_NAMES = {
    "scl_reg_kernel<10, 8, float>": "scl_reg_kernel<10>/L8",
    "scl_reg_kernel<10, 8, double>": "scl_reg_kernel<10,double>/L8",
    "sc_reg_full_kernel<double, 10>": "sc_reg_full_kernel<double,10>",
    "scl_wide_kernel<9, 32, float>": "scl_wide_kernel<9>/L32",
    "scl_wide_kernel<9, 32, double>": "scl_wide_kernel<9,double>/L32",
    "sc_big_kernel": "sc_big_kernel<14>",
    "sc_static_mc_kernel": "sc_static_mc_kernel<10>",
    "pbf_reg_kernel<float, 10, 0>": "pbf_reg_kernel<float,10>/T1",
}


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


def copy(src, dst_dir, name, required=True):
    path = src if os.path.isabs(src) else os.path.join(OUT, src)
    if not os.path.exists(path):
        if required:
            raise SystemExit("missing " + path)
        print("skipped (not there):", src)
        return False
    shutil.copy(path, os.path.join(dst_dir, name))
    return True


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "polar-codes_amd", "bench.py"], capture_output=True, text=True).stdout.strip()
    if dirty:
        print("warning: sources differ from HEAD, the recorded commit is not what was measured:\n" + dirty)
    date = datetime.date.today().isoformat()
    bench = json.loads(open(os.path.join(OUT, "bench.json")).read().strip().splitlines()[-1])
    kernel = bench["roofline"]["kernel"]
    frames = bench["config"]["frames_per_gpu"]
    short = kernel.split("<")[0]
    copy("bench.json", dst, "bench_n1.json")
    copy("tests_gpu.log", dst, "tests_gpu.log")
    copy(latest("prof_stats/*/*kernel_stats.csv"), dst, "bench_kernel_stats.csv")
    fpath, wpath = latest("pmcfull_FETCH_SIZE/*/*counter_collection.csv"), latest("pmcfull_WRITE_SIZE/*/*counter_collection.csv")
    copy(fpath, dst, "pmc_fetch_size.csv")
    copy(wpath, dst, "pmc_write_size.csv")
    fetch, nf = per_launch(fpath, "FETCH_SIZE", short)
    write, nw = per_launch(wpath, "WRITE_SIZE", short)
    per_frame = (2 * fetch + write) * 1024 / frames
    formula = "(2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 / frames per launch (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)"
    tab = {"_formula": formula, "_written_by": "tools/refresh_profiles.py " + tag}
    tab[kernel] = {"bytes_per_frame": per_frame, "what": "bench.py step: (1024,512) SC, %d frames per launch" % frames, "frames_per_launch": frames,
                   "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "file": "profiles/%s/pmc_fetch_size.csv + pmc_write_size.csv" % tag,
                   "commit": commit, "date": date}
    for name in ("legs_kernel_stats.csv", "legs_sq_counters.txt", "traffic_legs.json"):
        copy(name, dst, name)
    legs = json.load(open(os.path.join(OUT, "traffic_legs.json")))
    for prof_name, entry in legs.items():
        key = LEG_NAMES.get(prof_name)
        if key:
            tab[key] = {"bytes_per_frame": entry["bytes_per_frame"], "what": entry["what"], "frames_per_launch": entry["frames_per_launch"],
                        "FETCH_SIZE_KiB": entry["FETCH_SIZE_KiB"], "WRITE_SIZE_KiB": entry["WRITE_SIZE_KiB"],
                        "file": "profiles/%s/traffic_legs.json (scripts/gpu_prof_legs.sh)" % tag, "commit": commit, "date": date}
    json.dump(tab, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    copy("sc_sq_counters.txt", dst, "sc_sq_counters.txt", required=False)
    copy("scl_cycle_profile.txt", dst, "scl_cycle_profile.txt", required=False)
    copy("bench_n2_gloo.json", dst, "bench_n2_gloo_rehearsal.json", required=False)
    for name in ("scl_time.txt", "scl64_time.txt", "mc_time.txt"):
        copy(name, dst, name, required=False)
    print("commit %s: kernel %s: FETCH %.0f KiB (%d launches), WRITE %.0f KiB (%d launches) -> %.0f B per frame" % (commit, kernel, fetch, nf, write, nw, per_frame))
    print("bench: %.2f ms per step, %.0f M frames/s, kernel %.2f ms" % (bench["ms_per_step"], bench["frames_per_s"] / 1e6, bench["roofline"]["kernel_ms"]))
    for r in csv.DictReader(open(os.path.join(dst, "bench_kernel_stats.csv"))):
        if short in r["Name"]:
            print("rocprof: %s calls, average %.3f ms" % (r["Calls"], float(r["AverageNs"]) / 1e6))


if __name__ == "__main__":
    main()
