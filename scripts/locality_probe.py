"""Does spatial locality of concurrently rendered blocks matter?  Sum of the
per-block cycle counts of one image stripe when the whole frame is rendered vs
when only that stripe is (same blocks, same work, different cache company)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "items_loc.bin")
os.environ["MT_DEBUG_ITEM_CYCLES"] = out
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene()
abi.set_lights(h, sg.ROOM_LIGHTS)
abi.set_scheduling(h, False)  # material-classified launches: the same quarters in both runs
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
def dump(nx):
    a = np.fromfile(out, dtype=np.uint64).reshape(-1, 2)
    a = a[a[:, 0] > 0]
    item = ((a[:, 1] >> np.uint64(8)) & np.uint64(0xffffffff)).astype(np.int64)
    return a[:, 0].astype(np.float64), item % nx, item // nx
for _ in range(3):
    abi.render_chunk(h, sens, W, H)
d, bx, by = dump(W // 8)
for x0 in (0, 720, 840, 1440):
    sel = (bx * 8 >= x0) & (bx * 8 < x0 + 240)
    full = d[sel].sum()
    for _ in range(3):
        r = abi.render_chunk(h, sens, W, H, chunk=(x0, 0, 240, H))
    ds, _, _ = dump(240 // 8)
    print("stripe x0=%4d: cycles of its blocks in the full frame %.3e, alone %.3e (ratio %.3f), alone kernel_ms %.2f" % (x0, full, ds.sum(), ds.sum() / full, r["stats"]["kernel_ms"]))
    for _ in range(3):
        abi.render_chunk(h, sens, W, H)
