"""Why are the blocks at x=960 slow?  Primary rays of a few pixel columns."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene()
W, H = 1920, 1080
for x in (958, 959, 960, 961, 962, 1200):
    rays = np.array([binding.sensor_ray(sg.ROOM_CAMERA, W, H, x, y) for y in range(584, 648)])
    r = abi.intersect_rays(h, rays)
    st = r["stats"]
    print("x", x, "dir sample", rays[0, 3:], "zero comps", int((rays[:, 3:] == 0).sum()),
          {k: st[k] for k in ("box_tests", "node_visits", "tri_tests", "mt_tests")}, "kernel_ms %.3f" % st["kernel_ms"],
          "hit lines", sorted(set(r["line"].tolist()))[:6])
