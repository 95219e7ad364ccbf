"""A/B of traversal modes in one process: primary-only frame (coherent rays, one
pass per block) and the full 1080p frame, warm.  MODES=0,9 python scripts/ab.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
modes = [int(x) for x in os.environ.get("MODES", "0,9").split(",")]
abi = M.hip_abi()
for scene, (W, H) in (("room_nomtl", (1920, 1080)), ("room", (1920, 1080))):
    info = sg.write_scene(scene, "/tmp/mt_scenes")
    m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
    h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
    sens = binding.sensor(sg.ROOM_CAMERA, W, H)
    for rep in range(2):
        for mode in modes:
            abi.set_traversal_mode(h, mode)
            t = [abi.render_chunk(h, sens, W, H)["stats"]["kernel_ms"] for _ in range(6)]
            print("%s mode %d: warm min %.3f median %.3f ms" % (scene, mode, min(t[1:]), float(np.median(t[1:]))), flush=True)
            if scene == "room":  # plain blocks only (walls and ceiling: primary + 3 shadow passes each)
                t = [abi.render_chunk(h, sens, W, H, chunk=(0, 0, 952, 512))["stats"]["kernel_ms"] for _ in range(6)]
                print("%s left band 952x512 mode %d: warm min %.3f median %.3f ms" % (scene, mode, min(t[1:]), float(np.median(t[1:]))), flush=True)
