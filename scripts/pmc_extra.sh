#!/bin/bash
# Extra rocprofv3 counter groups for the bench's frame kernel (one --pmc group per pass, --kernel-trace only besides):
#   bash scripts/pmc_extra.sh <tag> "<kernel substring>" "<group 1>" "<group 2>" ...   (run through gpurun)
# prints per group the mean of every counter over the kernel's last 16 launches.
set -e
TAG=$1; KERNEL=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/x$i" --output-format csv -- python3 $R/bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-extras > "$OUT/x$i.log" 2>&1 || echo "pass $i ($grp) failed"
done
find "$OUT" -name "*.db" -delete 2>/dev/null || true
KERNEL="$KERNEL" OUT="$OUT" python3 - <<'PY'
import csv, glob, os, collections
out, kernel = os.environ["OUT"], os.environ["KERNEL"]
for d in sorted(glob.glob(out + "/x*")):
    if not os.path.isdir(d): continue
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(sorted(f, key=os.path.getmtime)[-1])):
        if kernel in row["Kernel_Name"]:
            acc[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    for name, per in acc.items():
        v = [per[k] for k in sorted(per)][-16:]
        print("%-32s %.5g  (mean of %d launches)" % (name, sum(v) / len(v), len(v)))
PY
