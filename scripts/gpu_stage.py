"""Staged GPU bring-up: each stage is tiny and prints before/after."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mythtracer_amd as M
import orclib
def say(*a):
    print(*a, flush=True)
stage = sys.argv[1]
CORNELL = os.path.join(ROOT, "tests", "scenes", "cornell_n.obj")
cam = (50, 50, -120, 0, 0, 0, 60); lights = [(50, 90, 50, .3, .3, .3, 1, 1, 1, 1, 1, 1)]
m = M.MythTracer(CORNELL); m.set_lights(lights)
o = orclib.OracleScene(CORNELL); o.set_lights(lights)
say("stage", stage, "devices", M.hip_abi().device_count())
if stage == "intersect":
    rays = np.array([[50, 50, -120, 0, 0, 1], [50, 50, -120, 0.1, 0.05, 0.99], [500, 500, 500, 0, 0, 1]], dtype=float)
    say("calling intersect"); g = m.intersect(rays); say("gpu", g["line"], g["t"])
    r = o.intersect(rays); say("cpu", r["line"], r["t"])
elif stage.startswith("render"):
    n = int(stage[6:] or 8)
    say("calling render", n); g = m.render(cam, 256, 256, chunk=(120, 120, n, n)); say("done", g["counters"], g["kernel_ms"])
    r = o.render(cam, 256, 256, chunk=(120, 120, n, n)); say("equal", np.array_equal(g["rgb"], r["rgb"]), r["counters"] == g["counters"])
    if not np.array_equal(g["rgb"], r["rgb"]): say(g["rgb"][0,:4], r["rgb"][0,:4])
