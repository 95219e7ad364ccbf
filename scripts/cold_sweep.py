"""First-frame (no cost history, ray-pool engine) time of the 1080p room frame against the eager-cut share."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); abi = M.hip_abi()
cold = []
for rep in range(3):
    h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS); sys.path.insert(0, "%s/scripts"); import knobs; knobs.from_env(abi, h)
    cold.append(abi.render_chunk(h, sens, W, H)["stats"]["kernel_ms"])
    abi.scene_destroy(h)
print("%%s: cold frames %%s" %% (os.environ.get("TAG"), " ".join("%%.2f" %% c for c in cold)), flush=True)
''' % (ROOT, ROOT)
for cut in os.environ.get("CUTS", "0.1,0.2,0.3,0.5,0.8").split(","):
    for cf in ("2.0", "3.0"):
        env = dict(os.environ, MT_DEBUG_CUT_SHARE=cut, MT_DEBUG_CELL_FACTOR=cf, TAG="cut %s cell factor %s" % (cut, cf))
        subprocess.run([sys.executable, "-c", code], env=env)
