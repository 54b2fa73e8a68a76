#!/bin/bash
# rt_headless with its options on a small moving scene (GPU box): bash scripts/cli_check.sh
set -e
cd "$GRAFT_REPO_ROOT"
X=gpuraytracer_amd/bin/rt_headless
COMMON="--synthetic 20000 --width 160 --height 90 --cycles 3 --dynamic"
timeout -k 5 60 $X $COMMON --out gpurun_out/cli_a.png --pfm gpurun_out/cli_a.pfm | tail -1
timeout -k 5 60 $X $COMMON --no-previews --out gpurun_out/cli_b.png --pfm gpurun_out/cli_b.pfm | tail -1
cmp gpurun_out/cli_a.pfm gpurun_out/cli_b.pfm && echo "previews off: same image"
timeout -k 5 60 $X $COMMON --refit --out gpurun_out/cli_c.png --pfm gpurun_out/cli_c.pfm | tail -1
cmp gpurun_out/cli_a.pfm gpurun_out/cli_c.pfm && echo "refit: same image (closest hits do not depend on the hierarchy, ties aside)" || echo "refit: image differs in tie cases"
timeout -k 5 60 $X --synthetic 20000 --width 160 --height 90 --cycles 2 --ranks 1 --drop-missed-rays --out gpurun_out/cli_d.png | tail -1
