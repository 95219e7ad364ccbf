"""Frame time of a launch WITHOUT cost history (first frame of a geometry)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
abi.set_scheduling(h, False)
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
ref = None
for _ in range(4):
    r = abi.render_chunk(h, sens, W, H)
a, b = abi.kernel_times(h)
print("%s: primary %.2f ms + render %.2f ms = %.2f ms" % (
    "launch without cost history", a[-1], b[-1], a[-1] + b[-1]))
