"""Summarises gpurun_out/prof_<tag>/ (scripts/pmc_passes.sh) for one kernel.

  python scripts/pmc_summary.py <tag> --kernel "render_kernel<false>" [--skip N] [--last M]
                                [--json profiles/pmc_frame_kernel.json --key <workload key> --commit <hash>]

Kernel durations come from the --kernel-trace --stats pass, counters from the --pmc passes (one group per pass);
figures are means over the kernel's launches [skip, skip + last) in launch order -- the same launches in every pass,
since every pass runs the same program.  With --json the derived figures are merged into that file under --key,
stamped with the SHA-256 of the kernel sources they were measured on (bench.py reports them only while the sources
are the same) and the commit."""
import argparse, collections, csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("--kernel", required=True)
ap.add_argument("--skip", type=int, default=0)
ap.add_argument("--last", type=int, default=10 ** 6)
ap.add_argument("--json")
ap.add_argument("--key")
ap.add_argument("--commit", default="")
args = ap.parse_args()
base = os.path.join(ROOT, "gpurun_out", "prof_" + args.tag)
SEL = slice(args.skip, args.skip + args.last)


def by_kernel(rows, value):
    per = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        per.setdefault((r["Kernel_Name"], int(r["Dispatch_Id"])), 0.0)
        per[(r["Kernel_Name"], int(r["Dispatch_Id"]))] += value(r)
    out = collections.defaultdict(list)
    for (k, _), v in per.items():
        out[k].append(v)
    return out


def newest_per_dir(pattern):
    """One file per pass directory: gpurun MERGES a call's output into gpurun_out/, so an earlier run of the same tag
    leaves its files (other process ids in their names) beside the new ones -- only the newest of each directory counts."""
    best = {}
    for f in glob.glob(pattern, recursive=True):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


dur = {}
for f in newest_per_dir(os.path.join(base, "stats", "**", "*kernel_trace.csv")):
    dur = by_kernel(list(csv.DictReader(open(f))), lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("kernel durations, rocprofv3 --kernel-trace (ms per launch, in launch order)")
ms = None
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if "mt::" not in k:
        continue
    print("  %s: %d launches, total %.3f ms, mean %.3f" % (k[:72], len(v), sum(v), sum(v) / len(v)))
    if args.kernel in k:
        t = v[SEL]
        ms = sum(t) / len(t)
        print("     launches [%d, %d): %s" % (args.skip, args.skip + len(t), " ".join("%.3f" % x for x in t)))
        print("     mean %.3f  min %.3f  max %.3f" % (ms, min(t), max(t)))
cnt = collections.defaultdict(dict)
for f in newest_per_dir(os.path.join(base, "pmc*", "**", "*counter_collection.csv")):
    rows = list(csv.DictReader(open(f)))
    for c in sorted(set(r["Counter_Name"] for r in rows)):
        per = by_kernel([r for r in rows if r["Counter_Name"] == c], lambda r: float(r["Counter_Value"]))
        for k, v in per.items():
            cnt[k][c] = v
derived = {}
for k in cnt:
    if args.kernel not in k:
        continue
    print("\ncounters of %s, mean over launches [%d, %d)" % (k[:60], args.skip, args.skip + args.last))
    m = {}
    for c, v in sorted(cnt[k].items()):
        t = v[SEL]
        m[c] = sum(t) / len(t)
        print("  %-26s %.6g" % (c, m[c]))
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        # MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
        # counts 64-B requests as 32 B (x2)
        rd, wr = m["FETCH_SIZE"] * 1024 * 2, m["WRITE_SIZE"] * 1024
        print("  memory-side traffic per launch: read %.1f MB  write %.1f MB  total %.1f MB" % (rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
        derived["hbm_bytes_per_launch"] = int(rd + wr)
    if "TCC_HIT_sum" in m:
        derived["l2_bytes_per_launch"] = int((m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) * 128)
        derived["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
        print("  L2: %.1f GB requested per launch, hit rate %.4f" % (derived["l2_bytes_per_launch"] / 1e9, derived["l2_hit_rate"]))
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_ANY" in m and "SQ_ACTIVE_INST_VALU" in m:
        derived["wave_time_share"] = {"valu_busy": m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"],
                                      "s_waitcnt": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
                                      "issue_stalled": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]}
        print("  share of wave time: VALU busy %.3f  waiting on s_waitcnt %.3f  issue-stalled %.3f" % tuple(derived["wave_time_share"].values()))
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if c in m:
            derived[c.lower() + "_per_launch"] = m[c]
    if ms and "SQ_ACTIVE_INST_VALU" in m:
        # SQ_ACTIVE_INST_VALU counts quad-cycles (4 shader cycles); 1024 SIMDs; the clock is GRBM_GUI_ACTIVE / 8 XCDs /
        # duration (MI355X_MICROARCH.md, DVFS section)
        clk = m.get("GRBM_GUI_ACTIVE", 0) / 8.0 / (ms * 1e-3) if m.get("GRBM_GUI_ACTIVE") else 2.4e9
        busy = m["SQ_ACTIVE_INST_VALU"] * 4.0
        frac = busy / (1024.0 * clk * ms * 1e-3)
        print("  VALU issue: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x %.2f GHz x %.3f ms) = %.3f of the chip's VALU issue cycles" % (clk / 1e9, ms, frac))
        derived.update({"valu_busy_simd_cycles_per_launch": busy, "valu_issue_frac_under_rocprof": frac,
                        "effective_clock_GHz": clk / 1e9, "kernel_ms_under_rocprof": ms,
                        "formula": "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x clock x kernel time), clock = GRBM_GUI_ACTIVE / 8 / kernel time"})
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
        derived["active_lanes_per_valu_instruction"] = m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"]
        print("  active lanes per VALU instruction: %.1f of 64" % derived["active_lanes_per_valu_instruction"])
        if "valu_issue_frac_under_rocprof" in derived:
            print("  lane-weighted VALU issue: %.3f" % (derived["valu_issue_frac_under_rocprof"] * derived["active_lanes_per_valu_instruction"] / 64.0))
    if "SQC_DCACHE_REQ" in m and m["SQC_DCACHE_REQ"]:
        derived["scalar_cache_miss_rate"] = m["SQC_DCACHE_MISSES"] / m["SQC_DCACHE_REQ"]
        print("  scalar data cache miss rate %.3f" % derived["scalar_cache_miss_rate"])
if args.json and derived:
    import bench
    derived["kernel"] = args.kernel
    derived["kernel_source_sha256"] = bench.kernel_source_sha256()
    derived["commit"] = args.commit
    derived["launches_averaged"] = "[%d, %d) of the kernel's launches in the profiled program" % (args.skip, args.skip + args.last)
    path = os.path.join(ROOT, args.json)
    allp = json.load(open(path)) if os.path.exists(path) else {"_comment": "per-launch PMC figures of the frame kernels (scripts/pmc_passes.sh + pmc_summary.py); bench.py reports an entry only while kernel_source_sha256 matches the sources it runs"}
    allp[args.key] = derived
    json.dump(allp, open(path, "w"), indent=1, sort_keys=True)
    print("\nwrote %s [%s]" % (args.json, args.key))
