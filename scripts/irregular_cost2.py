import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); flat = m.flatten()
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
for name, path in (("r01", os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_r01.so")), ("now", None)):
    abi = M.HipAbi(path) if path else M.hip_abi()
    h = abi.scene_create(flat); abi.set_lights(h, sg.ROOM_LIGHTS)
    for cname, chunk in (("column 960", (960, 0, 8, 1080)), ("column 952", (952, 0, 8, 1080)), ("cols 0..951", (0, 0, 952, 1080)), ("cols 968..1919", (968, 0, 952, 1080)), ("full", None)):
        t = [abi.render_chunk(h, sens, W, H, chunk=chunk)["stats"]["kernel_ms"] for _ in range(5)]
        print("%-4s %-16s warm %.3f ms" % (name, cname, min(t[1:])), flush=True)
