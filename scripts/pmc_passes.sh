#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box (run through gpurun):
#   bash scripts/pmc_passes.sh <tag>
# pass 0: --kernel-trace --stats ; passes 1..n: one --pmc group each (no other
# trace domains, as the pool requires).  Output: gpurun_out/prof_<tag>/...
set -e
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras"
timeout -k 10 150 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- $BENCH > "$OUT/stats.log" 2>&1
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
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" --output-format csv -- $BENCH > "$OUT/pmc$i.log" 2>&1 || echo "pass $i ($grp) failed" >> "$OUT/failed.txt"
  echo "pass $i done" >> "$OUT/progress.txt"
done
find "$OUT" -name "*.db" -delete 2>/dev/null || true
du -sh "$OUT" | tail -1
