#!/bin/bash
# rocprofv3 --kernel-trace --stats of a longer default-workload run (20 timed steps), so that the per-kernel
# average is dominated by the timed launches:  bash scripts/stats_pass.sh <tag>
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/stats_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/run" --output-format csv -- \
  python3 $ROOT/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/bench.json" 2> "$OUT/log.txt"
find "$OUT" -name "*.db" -delete 2>/dev/null || true
cat "$OUT"/run/*/*kernel_stats.csv | head -5
