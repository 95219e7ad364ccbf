import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "items.bin")
os.environ["MT_DEBUG_ITEM_CYCLES"] = out
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
W, H = 1920, 1080
n_frames = int(os.environ.get("FRAMES", "3"))  # frame 1 has no cost history, the later ones do
for _ in range(n_frames):
    g = m.render(sg.ROOM_CAMERA, W, H)
a = np.fromfile(out, dtype=np.uint64).reshape(-1, 2)
a = a[a[:, 0] > 0]
d = a[:, 0].astype(np.float64)
passes = (a[:, 1] >> np.uint64(40)).astype(np.float64)
item = ((a[:, 1] >> np.uint64(8)) & np.uint64(0xffffffff)).astype(np.int64); sub = (a[:, 1] & np.uint64(0xff)).astype(np.int64) - 1
steps = np.ones_like(passes)
bx = (item % (W // 8)) * 8; by = (item // (W // 8)) * 8
print("items", len(d), "kernel_ms", g["kernel_ms"])
print("ticks: sum %.3e mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f" % (d.sum(), d.mean(), np.median(d), np.percentile(d, 90), np.percentile(d, 99), d.max()))
print("passes: mean %.1f median %.0f p99 %.0f max %.0f" % (passes.mean(), np.median(passes), np.percentile(passes, 99), passes.max()))
order = np.argsort(-d)[:8]
for i in order: print(" work", i, "cycles %.3e passes %d item %d sub %d block x %d y %d" % (d[i], passes[i], item[i], sub[i], bx[i], by[i]))
print("quarter units: %d, whole blocks: %d" % ((sub >= 0).sum(), (sub < 0).sum()))
med = np.argsort(np.abs(d - np.median(d)))[:3]
for i in med: print(" median-ish item", i, "cycles %.3e passes %d" % (d[i], passes[i]))
c = np.sort(d)[::-1]
print("share of total cycles in top 1%% items: %.3f ; top 10%%: %.3f" % (c[:324].sum() / c.sum(), c[:3240].sum() / c.sum()))
# greedy list-scheduling simulation: how long would the frame take if work items
# were handed out in this order to 3072 / 2048 waves (no interference model)
import heapq
for nw in (2048, 3072):
    h = [0.0] * nw
    heapq.heapify(h)
    for x in d:
        t = heapq.heappop(h); heapq.heappush(h, t + x)
    mk = max(h)
    print("waves %d: makespan in given order %.3e cycles (%.1f ms @2.4GHz), perfect balance %.3e, max item %.3e" % (nw, mk, mk / 2.4e6, d.sum() / nw, d.max()))
    hs = [0.0] * nw; heapq.heapify(hs)
    for x in np.sort(d)[::-1]:
        t = heapq.heappop(hs); heapq.heappush(hs, t + x)
    print("   longest-first order: makespan %.3e (%.1f ms)" % (max(hs), max(hs) / 2.4e6))
