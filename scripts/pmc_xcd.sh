set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for sc in room loft; do for t in 0 2; do
  SCENE=$sc TUNE=XCD_QUEUES=$t timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $R/gpurun_out/r4q/pmc_${sc}_$t --output-format csv -- python3 $R/scripts/scene_frame.py > $R/gpurun_out/r4q/pmc_${sc}_$t.log 2>&1
done; done
find $R/gpurun_out/r4q -name "*.db" -delete
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for d in sorted(glob.glob(R+"/gpurun_out/r4q/pmc_*_?")):
    f=glob.glob(d+"/**/*counter_collection.csv",recursive=True)
    if not f: print(d,"no csv"); continue
    acc=collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        if "render_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    n=len(acc["TCC_HIT_sum"])
    h=sum(acc["TCC_HIT_sum"][-16:])/16; m=sum(acc["TCC_MISS_sum"][-16:])/16
    print(os.path.basename(d), "launches",n,"hit %.3e miss %.3e rate %.4f"%(h,m,h/(h+m)))
PY
