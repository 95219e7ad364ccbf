"""How much of the 1080p frame is the pixel column whose rays have a zero direction component (camera on the root's
split plane)?  The same frame with the camera moved off the plane by 1e-3."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
W, H = 1920, 1080
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
for name, cam in (("on the plane", sg.ROOM_CAMERA), ("off the plane", (200.001,) + tuple(sg.ROOM_CAMERA[1:])), ("yaw 1 degree", sg.ROOM_CAMERA[:4] + (1.0,) + sg.ROOM_CAMERA[5:])):
    sens = binding.sensor(cam, W, H)
    for engine in (1, 2):
        abi.set_engine(h, engine)
        t = [abi.render_chunk(h, sens, W, H)["stats"]["kernel_ms"] for _ in range(8)]
        print("%-14s engine %d: cold %.2f warm min %.3f median %.3f ms" % (name, engine, t[0], min(t[2:]), float(np.median(t[2:]))), flush=True)
