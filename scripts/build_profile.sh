#!/bin/bash
# Per-kernel table of the fused build alone: bash scripts/build_profile.sh <out-dir> <triangles> <second hierarchy 0|1>
set -u
OUT=$1; N=$2; SECOND=${3:-0}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 200 python scripts/build_time.py "$N" 20 "$SECOND" > "$OUT/time_${N}_${SECOND}.txt" 2>&1 || { tail -3 "$OUT/time_${N}_${SECOND}.txt"; exit 1; }
cat "$OUT/time_${N}_${SECOND}.txt"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${N}_${SECOND}" -o p -- python scripts/build_time.py "$N" 20 "$SECOND" 0 > /dev/null 2> "$OUT/rocprof_${N}_${SECOND}.err" || { tail -3 "$OUT/rocprof_${N}_${SECOND}.err"; exit 1; }
python - "$OUT/prof_${N}_${SECOND}/p_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name[:name.index("(")] if "(" in name else name
    per_build = float(r["TotalDurationNs"]) / 23.0 / 1e3      # 3 warm-up + 20 timed builds
    tot += per_build
    print("  %-44s %5d calls  avg %9.2f us   per build %9.2f us" % (name[:44], int(r["Calls"]), float(r["AverageNs"]) / 1e3, per_build))
print("  sum of kernel time per build: %.1f us" % tot)
PY
