"""Passes and rays per work unit of the 1080p room frame: how full is a pass?"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "items_fill.bin")
os.environ["MT_DEBUG_ITEM_CYCLES"] = out
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
W, H = 1920, 1080
for _ in range(3):
    g = m.render(sg.ROOM_CAMERA, W, H)
st = g["counters"]
rays = sum(int(st[k]) for k in ("rays_primary", "rays_secondary", "rays_shadow"))
a = np.fromfile(out, dtype=np.uint64).reshape(-1, 2)
a = a[a[:, 0] > 0]
d = a[:, 0].astype(np.float64)
passes = (a[:, 1] >> np.uint64(40)).astype(np.float64)
sub = (a[:, 1] & np.uint64(0xff)).astype(np.int64) - 1
print("kernel_ms", g.get("kernel_ms"), "rays", rays, {k: int(st[k]) for k in st if k.startswith("rays") or k in ("node_visits", "wave_node_steps", "wave_tri_steps", "mt_tests", "shaded_hits")})
print("units %d passes total %.0f -> rays per pass %.1f" % (len(d), passes.sum(), rays / passes.sum()))
print("cycles total %.3e -> cycles per pass %.0f ; longest unit %.3e ; perfect balance over 3072 waves %.3e" % (d.sum(), d.sum() / passes.sum(), d.max(), d.sum() / 3072))
for name, sel in (("whole blocks", sub < 0), ("quarters", (sub >= 0) & (sub < 4)), ("2x2 cells", sub >= 4)):
    if sel.sum():
        print(" %s: units %d passes %.0f cycles %.3e (%.1f%% of all) cycles/pass %.0f" % (
            name, sel.sum(), passes[sel].sum(), d[sel].sum(), 100 * d[sel].sum() / d.sum(), d[sel].sum() / passes[sel].sum()))
# histogram of passes per unit weighted by cycles
edges = [0, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 1e9]
for lo, hi in zip(edges[:-1], edges[1:]):
    sel = (passes > lo) & (passes <= hi)
    if sel.sum():
        print(" passes in (%g,%g]: units %d, cycles share %.3f, cycles/pass %.0f" % (lo, hi, sel.sum(), d[sel].sum() / d.sum(), d[sel].sum() / passes[sel].sum()))
