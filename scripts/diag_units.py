"""Per-unit step counts of the traversal (diagnostic -DMT_DIAG build): what does a
pass of a heavy unit consist of, compared with a pass of a plain one?"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "items_diag.bin")
os.environ["MT_DEBUG_ITEM_CYCLES"] = out
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
abi = M.HipAbi(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_diag.so"))
h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS)
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
abi.set_traversal_mode(h, int(os.environ.get("MODE", "0")))
for _ in range(3):
    r = abi.render_chunk(h, sens, W, H)
a = np.fromfile(out, dtype=np.uint64).reshape(-1, 2)
n = a.shape[0] // 3
u, d, tt = a[:n], a[n:2 * n], a[2 * n:]
ok = u[:, 0] > 0
u, d, tt = u[ok], d[ok], tt[ok]
trace_ticks = tt[:, 0].astype(np.float64)
cyc = u[:, 0].astype(np.float64)
passes = (u[:, 1] >> np.uint64(40)).astype(np.float64)
sub = (u[:, 1] & np.uint64(0xff)).astype(np.int64) - 1
nodes = (d[:, 0] >> np.uint64(32)).astype(np.float64)
atrips = (d[:, 0] & np.uint64(0xffffffff)).astype(np.float64)
rays = (d[:, 1] >> np.uint64(40)).astype(np.float64)
trans = ((d[:, 1] >> np.uint64(20)) & np.uint64(0xfffff)).astype(np.float64)
vec = (d[:, 1] & np.uint64(0xfffff)).astype(np.float64)
print("kernel_ms", r["stats"]["kernel_ms"], "units", len(cyc))
def line(name, sel):
    p = passes[sel].sum()
    print("%-28s units %6d passes %7.0f | per pass: outside trace %6.0f cycles %7.0f rays %5.1f big steps %6.1f (vec %5.1f transposed %5.1f) A trips %5.1f | cycles share %.3f" % (
        name, sel.sum(), p, (cyc[sel].sum() - trace_ticks[sel].sum()) / p, cyc[sel].sum() / p, rays[sel].sum() / p, nodes[sel].sum() / p, vec[sel].sum() / p,
        trans[sel].sum() / p, atrips[sel].sum() / p, cyc[sel].sum() / cyc.sum()))
line("all", passes > 0)
line("whole, <= 4 passes", (sub < 0) & (passes <= 4))
line("whole, 5..8 passes", (sub < 0) & (passes > 4) & (passes <= 8))
line("whole, 9..16 passes", (sub < 0) & (passes > 8) & (passes <= 16))
line("whole, > 16 passes", (sub < 0) & (passes > 16))
line("quarters", (sub >= 0) & (sub < 4))
if (sub >= 4).any(): line("cells", sub >= 4)
cpp = cyc / passes
for lo, hi in ((0, 2e5), (2e5, 3e5), (3e5, 5e5), (5e5, 8e5), (8e5, 2e7)):
    line("cycles/pass in [%.0e,%.0e)" % (lo, hi), (cpp >= lo) & (cpp < hi))
print("heaviest units:")
for i in np.argsort(-cyc)[:12]:
    p = passes[i]
    print("  cycles %.3e passes %3d sub %2d | per pass: cycles %8.0f rays %5.1f big steps %6.1f (vec %5.1f transposed %5.1f) A trips %6.1f" % (
        cyc[i], p, sub[i], cyc[i] / p, rays[i] / p, nodes[i] / p, vec[i] / p, trans[i] / p, atrips[i] / p))
