"""Summarises gpurun_out/prof_<tag>/ (scripts/pmc_passes.sh) per kernel.

bench.py --steps 4 --warmup 2 --no-extras launches, in this order: warm-up 1 (no
cost history: probe + pool_kernel<true>), warm-up 2 and the counting frame
(render_kernel<true>), then the kernels built WITHOUT the work counters: one
first launch and the four timed steps (render_kernel<false>), then 16 more frames
with the counters.  Figures below are means over the four TIMED launches =
launches 2..5 of render_kernel<false>; the other launches are listed for
reference.
With a second argument the derived figures are also written as JSON (the entry
bench.py reads from profiles/hbm_traffic.json).

Usage: python scripts/pmc_summary.py <tag> > profiles/<tag>_pmc_summary.txt"""
import csv, glob, os, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
TIMED = slice(1, 5)   # of render_kernel<false>
FRAME_KERNEL = "render_kernel<false>"


def by_kernel(rows, value):
    """rows of one pass -> {kernel: [value per dispatch, in launch order]}"""
    per = collections.OrderedDict()
    for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
        per.setdefault((r["Kernel_Name"], int(r["Dispatch_Id"])), 0.0)
        per[(r["Kernel_Name"], int(r["Dispatch_Id"]))] += value(r)
    out = collections.defaultdict(list)
    for (k, _), v in per.items():
        out[k].append(v)
    return out


dur = {}
for f in glob.glob(os.path.join(base, "stats", "**", "*kernel_trace.csv"), recursive=True):
    dur = by_kernel(list(csv.DictReader(open(f))),
                    lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("kernel durations, rocprofv3 --kernel-trace (ms per launch, in launch order)")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if not k.startswith(("void mt::", "mt::")):
        continue
    print("  %s" % k[:70])
    print("     all launches: %s" % " ".join("%.3f" % x for x in v))
    if FRAME_KERNEL in k and len(v) >= 5:
        t = v[TIMED]
        print("     timed launches (2..5): mean %.3f  min %.3f  max %.3f" % (sum(t) / len(t), min(t), max(t)))
cnt = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(base, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    for c in sorted(set(r["Counter_Name"] for r in rows)):
        per = by_kernel([r for r in rows if r["Counter_Name"] == c], lambda r: float(r["Counter_Value"]))
        for k, v in per.items():
            cnt[k][c] = v
for k in cnt:
    if FRAME_KERNEL not in k:
        continue
    print("\ncounters of %s, mean over the timed launches (2..5)" % k[:60])
    m = {}
    for c, v in sorted(cnt[k].items()):
        t = v[TIMED] if len(v) >= 5 else v
        m[c] = sum(t) / len(t)
        print("  %-26s %.6g" % (c, m[c]))
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        # MI355X_MICROARCH.md, HBM / rocprofv3 section: FETCH_SIZE and WRITE_SIZE are in
        # KiB; on gfx950 FETCH_SIZE counts 64-B requests as 32 B (x2)
        rd, wr = m["FETCH_SIZE"] * 1024 * 2, m["WRITE_SIZE"] * 1024
        print("  memory-side traffic per launch: read %.1f MB  write %.1f MB  total %.1f MB" % (rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
        print("  HBM_TRAFFIC_BYTES %d" % int(rd + wr))
    if "TCC_HIT_sum" in m:
        print("  L2 hit rate %.4f" % (m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])))
    derived = {}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        derived["hbm_bytes_per_launch"] = int(m["FETCH_SIZE"] * 1024 * 2 + m["WRITE_SIZE"] * 1024)
    if "SQ_WAVE_CYCLES" in m:
        print("  share of wave time: VALU busy %.3f  waiting on s_waitcnt %.3f  issue-stalled %.3f" % (
            m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"], m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
            m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
        derived["wave_time_share"] = {"valu_busy": m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"],
                                      "s_waitcnt": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"],
                                      "issue_stalled": m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]}
    t = dur.get(k)
    if t and len(t) >= 5 and "SQ_ACTIVE_INST_VALU" in m:
        ms = sum(t[TIMED]) / len(t[TIMED])
        # SQ_ACTIVE_INST_VALU counts quad-cycles (4 shader cycles); 1024 SIMDs; the clock is
        # GRBM_GUI_ACTIVE / 8 XCDs / duration (MI355X_MICROARCH.md, DVFS section)
        clk = m.get("GRBM_GUI_ACTIVE", 0) / 8.0 / (ms * 1e-3) if m.get("GRBM_GUI_ACTIVE") else 2.4e9
        frac = m["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * clk * ms * 1e-3)
        print("  VALU issue: SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x %.2f GHz x %.3f ms) = %.3f of the chip's VALU issue cycles" % (clk / 1e9, ms, frac))
        derived["name"] = "VALU issue"
        derived["frac"] = frac
        derived["formula"] = "SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x clock x kernel time), clock = GRBM_GUI_ACTIVE / 8 / kernel time"
        derived["effective_clock_GHz"] = clk / 1e9
        derived["kernel_ms_under_rocprof"] = ms
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m:
        lanes = m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"]
        print("  active lanes per VALU instruction: %.1f of 64" % lanes)
        derived["active_lanes_per_valu_instruction"] = lanes
    if "TCC_HIT_sum" in m:
        l2 = (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) * 128
        print("  L2 requests x 128 B = %.1f GB per launch" % (l2 / 1e9))
        derived["l2_bytes_per_launch"] = int(l2)
        derived["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if "SQC_DCACHE_REQ" in m and m["SQC_DCACHE_REQ"]:
        derived["scalar_cache_miss_rate"] = m["SQC_DCACHE_MISSES"] / m["SQC_DCACHE_REQ"]
        print("  scalar data cache miss rate %.3f" % derived["scalar_cache_miss_rate"])
    if len(sys.argv) > 2 and derived:
        import json
        json.dump(derived, open(sys.argv[2], "w"), indent=1, sort_keys=True)
