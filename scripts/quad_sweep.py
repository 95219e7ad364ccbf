"""Sweep of the state machine's cutting constants on the warm 1080p room frame (MT_DEBUG_QUAD_SHARE / _WORK)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
abi.set_engine(h, 1)
for work, share, keep in [(w, s_, k) for w in (2.0, 2.4) for s_ in (0.9, 1.0, 1.1) for k in (0.5, 0.7, 0.85)]:
    if True:
        os.environ["MT_DEBUG_QUAD_SHARE"] = str(share); os.environ["MT_DEBUG_QUAD_WORK"] = str(work); os.environ["MT_DEBUG_QUAD_KEEP"] = str(keep)
        t = [abi.render_chunk(h, sens, W, H)["stats"]["kernel_ms"] for _ in range(9)]
        print("quad_work %.1f quad_share %.2f keep %.2f: frames %s -> min %.3f median %.3f" % (work, share, keep, " ".join("%.2f" % x for x in t[1:]), min(t[1:]), float(np.median(t[1:]))), flush=True)
