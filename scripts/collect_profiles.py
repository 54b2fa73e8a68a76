"""Copies the evidence of one GPU run (bench lines, rocprofv3 kernel table, counter passes) from a gpurun_out/
directory into profiles/ under the names the documentation cites.
Usage: python scripts/collect_profiles.py gpurun_out/<dir> [state line]"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def main():
    src = sys.argv[1].rstrip("/") + "/"
    state = sys.argv[2] if len(sys.argv) > 2 else ""
    prof = os.path.join(ROOT, "profiles")
    rows = list(csv.DictReader(open(src + "prof/p_kernel_stats.csv")))
    out = ["rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline   (MI355X, C3 workload)",
           "the run holds 7 timed/warm-up frames by k_traverse_spatial<false> (+ its second launch of k_traverse_quad for listed rays: none "
           "here), 3 frames by k_traverse_quad alone (the walk in the reference's order, reported beside the headline), 1 frame by the "
           "counting variant k_traverse_spatial<true> and 1 frame counting the reference's tests (k_traverse<true>); raw table: "
           "r1_c3_spatial_kernel_stats.csv",
           "k_traverse_quad's calls are 32 launches that find an empty list and return (a few us each) and 12 full walks of 61.5 ms: "
           "the average below mixes them",
           "state: " + state, "",
           "%-36s %6s %14s %14s %8s" % ("kernel", "calls", "total ms", "average us", "percent")]
    for r in rows:
        out.append("%-36s %6d %14.3f %14.3f %8.3f" % (short(r["Name"])[:60], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                     float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    open(os.path.join(prof, "r1_c3_spatial_kernel_stats_summary.txt"), "w").write("\n".join(out) + "\n")
    shutil.copy(src + "prof/p_kernel_stats.csv", os.path.join(prof, "r1_c3_spatial_kernel_stats.csv"))
    for a, b in (("bench.json", "r1_c3_spatial_bench.json"), ("bench_under_rocprof.json", "r1_c3_spatial_bench_under_rocprof.json"),
                 ("c4.json", "r1_c4_spatial_bench.json"), ("c5.json", "r1_c5_spatial_bench.json")):
        if os.path.exists(src + a) and os.path.getsize(src + a):
            shutil.copy(src + a, os.path.join(prof, b))
            d = json.load(open(src + a))
            print(a, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["lbvh_build_ms"], d.get("reference_order_walk"),
                  d["roofline"].get("own_layout", {}).get("steps_per_ray"), d["roofline"].get("own_layout", {}).get("achieved"))
    if os.path.isdir(src + "pmc"):
        text = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "pmc_summary.py"), src + "pmc"], capture_output=True, text=True).stdout
        open(os.path.join(prof, "r1_c3_spatial_pmc_summary.txt"), "w").write(text)
        on = False
        for line in text.split("\n"):
            if line and not line.startswith(" "):
                on = line.startswith("k_traverse_spatial<false>")
            elif on and any(k in line for k in ("FETCH_SIZE", "WRITE_SIZE", "GRBM", "ACTIVE_INST_VALU", "THREAD_CYCLES", "TA_BUSY", "TOTAL_CACHE")):
                print(line)
    for r in rows[:6]:
        print("%-36s %6d %12.3f us" % (short(r["Name"])[:36], int(r["Calls"]), float(r["AverageNs"]) / 1e3))


if __name__ == "__main__":
    main()
