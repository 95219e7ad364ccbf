import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
W, H = 1920, 1080
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); abi = M.hip_abi(); h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS)
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
abi.set_engine(h, int(os.environ.get("ENGINE", "1")))
t = []
for _ in range(5):
    abi.set_scheduling(h, True)  # forget the costs: every frame is a first frame
    t.append(abi.render_chunk(h, sens, W, H)["stats"]["kernel_ms"])
print("engine %s cold frames: %s" % (os.environ.get("ENGINE", "1"), " ".join("%.2f" % x for x in t)))
