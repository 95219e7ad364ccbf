"""Traversal modes on clean workloads: depth-0 frame (primary + shadow passes only) and the full frame."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); flat = m.flatten()
abi = M.hip_abi(); h = abi.scene_create(flat); abi.set_lights(h, sg.ROOM_LIGHTS)
for mode in [int(x) for x in os.environ.get("MODES", "0,8,9,10,11").split(",")]:
    abi.set_traversal_mode(h, mode)
    for cname, chunk, depth in (("depth 0 left 952", (0, 0, 952, 1080), 0), ("depth 5 left 952", (0, 0, 952, 1080), 5)):
        t = [abi.render_chunk(h, sens, W, H, chunk=chunk, max_depth=depth)["stats"]["kernel_ms"] for _ in range(6)]
        print("mode %2d %-18s warm min %.3f median %.3f ms" % (mode, cname, min(t[1:]), float(np.median(t[1:]))), flush=True)
