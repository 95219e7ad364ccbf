"""Cost of the pixel column whose rays have d.x == 0 exactly (camera on the root's split plane)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
for mode in [int(x) for x in os.environ.get("MODES", "0,10,1").split(",")]:
    abi.set_traversal_mode(h, mode)
    for name, chunk in (("column 960 (8 wide)", (960, 0, 8, 1080)), ("column 952", (952, 0, 8, 1080)), ("column 1200", (1200, 0, 8, 1080))):
        t = []
        for _ in range(4):
            r = abi.render_chunk(h, sens, W, H, chunk=chunk)
            t.append(r["stats"]["kernel_ms"])
        st = r["stats"]
        print("mode %2d %-20s warm %.3f ms rays %d node_visits %d" % (mode, name, min(t[1:]), st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"], st["node_visits"]), flush=True)
