"""Quick GPU sanity run: product (HIP) vs oracle on a few scenes."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mythtracer_amd as M
import orclib
from mythtracer_amd import scenegen as sg

out = os.path.join(ROOT, "gpurun_out"); os.makedirs(out, exist_ok=True)
scenes = "/tmp/mt_scenes"
for n in ["mini", "mini_nomtl", "room", "room_nomtl"]:
    sg.write_scene(n, scenes)
CORNELL = os.path.join(ROOT, "tests", "scenes", "cornell_n.obj")
cases = [("cornell", CORNELL, (50, 50, -120, 0, 0, 0, 60), [(50, 90, 50, .3, .3, .3, 1, 1, 1, 1, 1, 1)], 256, 256),
         ("mini", scenes + "/mini.obj", sg.ROOM_CAMERA, sg.ROOM_LIGHTS, 320, 180),
         ("mini_nomtl", scenes + "/mini_nomtl.obj", sg.ROOM_CAMERA, sg.ROOM_LIGHTS, 320, 180),
         ("room", scenes + "/room.obj", sg.ROOM_CAMERA, sg.ROOM_LIGHTS, 480, 270),
         ("room_nomtl", scenes + "/room_nomtl.obj", sg.ROOM_CAMERA, sg.ROOM_LIGHTS, 480, 270)]
only = sys.argv[1:] 
for name, obj, cam, lights, W, H in cases:
    if only and name not in only: continue
    m = M.MythTracer(obj)
    m.set_lights(lights)
    t0 = time.time(); g = m.render(cam, W, H, debug=True); t1 = time.time()
    g2 = m.render(cam, W, H)
    o = orclib.OracleScene(obj); o.set_lights(lights)
    r = o.render(cam, W, H, debug=True)
    diff = (g["rgb"].astype(int) - r["rgb"].astype(int))
    nd = int((diff != 0).any(axis=2).sum())
    print(name, "pixels differing:", nd, "max abs", int(np.abs(diff).max()),
          "line eq", bool(np.array_equal(g["line"], r["line"])),
          "point eq", bool(np.array_equal(g["point"], r["point"], equal_nan=True)),
          "counters eq", g["counters"] == r["counters"], flush=True)
    if g["counters"] != r["counters"]:
        print("  gpu", g["counters"]); print("  cpu", r["counters"])
    rays = sum(g["counters"][k] for k in ("rays_primary", "rays_secondary", "rays_shadow"))
    print("  kernel_ms %.3f (2nd %.3f) total_ms %.3f rays %d -> %.2f Mray/s; oracle %.3fs" % (
        g["kernel_ms"], g2["kernel_ms"], g["total_ms"], rays, rays / g2["kernel_ms"] / 1e3, r["seconds"]), flush=True)
    if nd:
        ys, xs = np.nonzero((diff != 0).any(axis=2))
        print("  first diffs:", [(int(x), int(y), g["rgb"][y, x].tolist(), r["rgb"][y, x].tolist()) for x, y in list(zip(xs, ys))[:5]])
    np.save(os.path.join(out, "quick_%s_gpu.npy" % name), g["rgb"])
