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
m = M.MythTracer(info["obj"]); abi = M.hip_abi(); h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS); sys.path.insert(0, "%s/scripts"); import knobs; knobs.from_env(abi, h)
t = [abi.render_chunk(h, sens, W, H)["stats"]["kernel_ms"] for _ in range(8)]
print("%%s: cold %%.2f warm min %%.3f median %%.3f" %% (os.environ.get("TAG"), t[0], min(t[2:]), float(np.median(t[2:]))))
''' % ROOT
for cut in ("0.8", "1.0", "1.2"):
    for pw in ("1.1,3.0", "2.0,3.0"):
        for pt in ("0.35,0.12", "0.5,0.12"):
            env = dict(os.environ, MT_ENGINE="2", MT_DEBUG_CUT_SHARE=cut, MT_DEBUG_PIECE_WORK=pw, MT_DEBUG_PIECE_TIME=pt,
                       TAG="cut %s work %s time %s" % (cut, pw, pt))
            subprocess.run([sys.executable, "-c", code], env=env)
