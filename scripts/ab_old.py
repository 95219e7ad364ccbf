"""Round-1 library vs the current one on identical workloads (one process, same scene object)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
libs = {"r01": os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_r01.so"), "mix": os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_mix.so"), "now": None}
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
flat = m.flatten()
cases = [("full frame depth 5", None, 5), ("full frame depth 0", None, 0), ("left 952x1080 depth 0", (0, 0, 952, 1080), 0),
         ("left 952x1080 depth 5", (0, 0, 952, 1080), 5), ("top 1920x300 depth 5", (0, 0, 1920, 300), 5)]
for name, path in libs.items():
    if path and not os.path.exists(path):
        continue
    abi = M.HipAbi(path) if path else M.hip_abi()
    h = abi.scene_create(flat); abi.set_lights(h, sg.ROOM_LIGHTS)
    for cname, chunk, depth in cases:
        t = [abi.render_chunk(h, sens, W, H, chunk=chunk, max_depth=depth)["stats"]["kernel_ms"] for _ in range(6)]
        print("%-4s %-26s cold %.2f warm min %.3f median %.3f ms" % (name, cname, t[0], min(t[1:]), float(np.median(t[1:]))), flush=True)
