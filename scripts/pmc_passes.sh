#!/bin/bash
# rocprofv3 passes on the GPU box (run through gpurun):
#   bash scripts/pmc_passes.sh <tag> <program and arguments, python3 first>
# e.g.  bash scripts/pmc_passes.sh r03 python3 bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-extras
#       bash scripts/pmc_passes.sh r03_cold python3 scripts/profile_run.py cold
# pass 0: --kernel-trace --stats ; passes 1..n: one --pmc group each (no other trace domains, as the pool requires;
# the program itself follows `--` directly).  Output: gpurun_out/prof_<tag>/...
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
PROG="$1"; shift
ARGS=()
for a in "$@"; do case "$a" in /*|-*) ARGS+=("$a");; *) if [ -e "$ROOT/$a" ]; then ARGS+=("$ROOT/$a"); else ARGS+=("$a"); fi;; esac; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- $PROG "${ARGS[@]}" > "$OUT/stats.log" 2>&1
echo "stats done" >> "$OUT/progress.txt"
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" --output-format csv -- $PROG "${ARGS[@]}" > "$OUT/pmc$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/failed.txt"
  echo "pass $i done" >> "$OUT/progress.txt"
done
find "$OUT" -name "*.db" -delete 2>/dev/null || true
du -sh "$OUT" | tail -1
