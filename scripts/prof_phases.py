"""Phase profile of the traversal (diagnostic -DMT_PROF build)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
flat = m.flatten()
abi = M.HipAbi(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_prof.so"))
h = abi.scene_create(flat)
abi.set_lights(h, sg.ROOM_LIGHTS)
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
for chunk in [None, (816, 632, 16, 16)]:
    r = abi.render_chunk(h, sens, W, H, chunk=chunk)
    print("chunk", chunk, "kernel_ms", r["stats"]["kernel_ms"], flush=True)
