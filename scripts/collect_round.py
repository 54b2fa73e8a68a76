"""Copies the evidence of one scripts/profile_round.sh run from gpurun_out/ into profiles/ under per-round names:
<round>_<workload>_bench.json, _bench_under_rocprof.json, _kernel_stats.csv, _kernel_stats_summary.txt,
_pmc_summary.txt, _pmc.json (the file bench.py reads its counters from).
Usage: python scripts/collect_round.py gpurun_out/<dir> <round> <workload> [note]"""
import csv
import json
import os
import shutil
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
    src, rnd, wl = sys.argv[1].rstrip("/") + "/", sys.argv[2], sys.argv[3]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    prof = os.path.join(ROOT, "profiles")
    pre = os.path.join(prof, "%s_%s_" % (rnd, wl))
    sys.path.insert(0, ROOT)
    from bench import source_sha
    import subprocess
    head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "gpuraytracer_amd/csrc", "include"], cwd=ROOT, capture_output=True,
                                text=True).stdout.strip())
    stamp = "state: git %s%s, kernel sources %s" % (head[:12], " + uncommitted changes to the kernel sources" if dirty else "", source_sha())
    stats = src + "prof/p_kernel_stats.csv"
    if os.path.exists(stats):
        rows = list(csv.DictReader(open(stats)))
        out = ["rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --workload %s --steps 5 --warmup 2 --no-cpu-baseline   (MI355X)" % wl,
               "besides the 7 timed / warm-up frames the run holds the frames bench.py adds for its side figures: 3 frames traced by "
               "k_traverse_quad alone (the walk in the reference's order), 1 frame by the counting variant k_traverse_spatial<true> and "
               "1 frame counting the reference's tests (k_traverse<true, false>); k_traverse_quad is also launched after every "
               "k_traverse_spatial for the listed rays and returns at once when there are none, so its average mixes both",
               note, stamp, "",
               "%-44s %6s %14s %14s %8s" % ("kernel", "calls", "total ms", "average us", "percent")]
        for r in rows:
            out.append("%-44s %6d %14.3f %14.3f %8.3f" % (short(r["Name"])[:44], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                         float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
        open(pre + "kernel_stats_summary.txt", "w").write("\n".join(out) + "\n")
        shutil.copy(stats, pre + "kernel_stats.csv")
        for r in rows[:8]:
            print("%-44s %6d %12.3f us" % (short(r["Name"])[:44], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
    for a, b in (("bench.json", "bench.json"), ("bench_under_rocprof.json", "bench_under_rocprof.json"), ("pmc_summary.txt", "pmc_summary.txt"),
                 ("pmc.json", "pmc.json")):
        if os.path.exists(src + a) and os.path.getsize(src + a):
            shutil.copy(src + a, pre + b)
    if os.path.exists(pre + "pmc.json"):
        table = json.load(open(pre + "pmc.json"))
        if table.get("source_sha") != source_sha():
            print("WARNING: the counter passes were taken on other kernel sources (%s) than this tree's (%s)" % (table.get("source_sha"), source_sha()))
        table["head"] = head + ("+" if dirty else "")
        json.dump(table, open(pre + "pmc.json", "w"), indent=1, sort_keys=True)
    if os.path.exists(pre + "pmc_summary.txt"):
        text = open(pre + "pmc_summary.txt").read()
        open(pre + "pmc_summary.txt", "w").write(stamp + "\n" + text)
    if os.path.exists(src + "bench.json"):
        d = json.load(open(src + "bench.json"))
        print("bench:", d["value"], d["unit"], "ms/step", d["ms_per_step"], "traverse launch ms", d["roofline"]["avg_launch_ms"], "build ms",
              d["lbvh_build_ms"], "roofline frac", d["roofline"]["frac"], "vector", (d["roofline"].get("vector") or {}).get("frac"))


if __name__ == "__main__":
    main()
