"""Summarises gpurun_out/prof_<tag>/ (scripts/pmc_passes.sh) per kernel:
average duration from the kernel trace and per-launch averages of every
counter.  Usage: python scripts/pmc_summary.py <tag> > profiles/<tag>_pmc_summary.txt"""
import csv, glob, os, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(base, "stats", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("kernel durations (rocprofv3 --kernel-trace, ms): name  launches  mean  min  max")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("  %-60s %3d  %.3f  %.3f  %.3f" % (k[:60], len(v), sum(v) / len(v), min(v), max(v)))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(base, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        cnt[k][c].append(v)
for k in sorted(cnt, key=lambda k: -sum(dur.get(k, [0]))):
    print("\ncounters per launch (mean over launches): %s" % k[:80])
    for c in sorted(cnt[k]):
        v = cnt[k][c]
        print("  %-26s %.6g   (%d launches)" % (c, sum(v) / len(v), len(v)))
