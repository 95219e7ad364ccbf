import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
h = m.device_scene(); abi = M.hip_abi(); abi.set_lights(h, sg.ROOM_LIGHTS)
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
for mode in [0, 4, 3, 2, 1, 0]:
    abi.set_traversal_mode(h, mode)
    r = abi.render_chunk(h, sens, W, H)
    r = abi.render_chunk(h, sens, W, H)
    print("mode", mode, "kernel_ms %.2f" % r["stats"]["kernel_ms"], flush=True)
