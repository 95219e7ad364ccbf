"""Animation regime: the camera turns 2 degrees per frame (main_local.cc:51-76)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
W, H = 1920, 1080
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); abi = M.hip_abi(); h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS)
for label in ("static", "moving"):
    t = []
    for f in range(14):
        cam = list(sg.ROOM_CAMERA)
        if label == "moving":
            cam[4] += 2.0 * f
        r = abi.render_chunk(h, binding.sensor(cam, W, H), W, H)
        t.append(r["stats"]["kernel_ms"])
    print("%s: first %.2f then mean %.2f min %.2f max %.2f ms" % (label, t[0], np.mean(t[2:]), min(t[2:]), max(t[2:])), flush=True)
