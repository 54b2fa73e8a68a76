#!/bin/bash
# One GPU call's worth of evidence for a round: the bench line, the same command under rocprofv3 --kernel-trace --stats,
# and the hardware-counter passes (one rocprofv3 --pmc run per counter group). Usage (GPU box, from the repo root):
#   bash scripts/profile_round.sh <out-dir under gpurun_out/> [workload]
# Afterwards, in the build container: python scripts/collect_round.py <out-dir> <round> <workload>
set -u
OUT=$1; WL=${2:-c3}
# the side figures that walk the reference's hierarchy take minutes at 50 M triangles
EXTRA=""; if [ "$WL" = "c5" ]; then EXTRA="--no-counters --no-reference-walk --no-cpu-baseline --steps 3 --warmup 1"; fi
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python bench.py --workload "$WL" --no-others $EXTRA > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
echo "bench ok: $(cut -c1-160 "$OUT/bench.json")"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -o p -- python bench.py --workload "$WL" --steps 5 --warmup 2 --no-cpu-baseline --no-others $EXTRA \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err" || { echo "rocprof run failed"; tail -5 "$OUT/rocprof.err"; exit 1; }
echo "kernel trace ok"
bash scripts/pmc_passes.sh "$OUT/pmc" --workload "$WL" --no-counters || exit 1
python scripts/pmc_summary.py "$OUT/pmc" --json "$OUT/pmc.json" > "$OUT/pmc_summary.txt"
echo "counter passes ok"
