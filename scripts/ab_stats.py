"""Work counters of library variants on the 1080p room frame (see ab_lib.py)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
names = sys.argv[1:] + ["now"]
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
flat = m.flatten()
for name in names:
    path = None if name == "now" else os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_%s.so" % name)
    abi = M.HipAbi(path) if path else M.hip_abi()
    h = abi.scene_create(flat); abi.set_lights(h, sg.ROOM_LIGHTS)
    for engine in (1, 2):
        abi.set_engine(h, engine)
        for depth in (5, 0):
            for _ in range(3):
                r = abi.render_chunk(h, sens, W, H, max_depth=depth)
            print(name, "engine", engine, "depth", depth, {k: (int(v) if isinstance(v, (int, np.integer)) else round(float(v), 3)) for k, v in r["stats"].items()}, flush=True)
    abi.lib.mt_scene_destroy(h)
