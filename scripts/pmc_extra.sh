#!/bin/bash
# extra one-off PMC groups: bash scripts/pmc_extra.sh <tag> "<counters>" ["<counters>" ...]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras"
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" --output-format csv -- $BENCH > "$OUT/pmc$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/failed.txt"
  echo "pass $i done" >> "$OUT/progress.txt"
done
