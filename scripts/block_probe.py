"""What does ONE 8x8 block of the 3840x2160 frame cost by itself?  Renders the blocks given as x,y pairs (pan step
STEP of the bench's camera) as chunks of their own, with the work counters on, through either engine.

  STEP=3 BLOCKS=1808,1256:1848,1240:3112,1072 python scripts/block_probe.py"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
abi = M.HipAbi(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_%s.so" % os.environ["LIB"])) if os.environ.get("LIB") else M.hip_abi()
h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS)
W, H = 3840, 2160
j = int(os.environ.get("STEP", "3")) % 16
tri = j if j <= 4 else (8 - j if j <= 12 else j - 16)
c = list(sg.ROOM_CAMERA); c[4] += 2.0 * tri
sens = binding.sensor(c, W, H)
if os.environ.get("TRAV"): abi.set_traversal_mode(h, int(os.environ["TRAV"]))
for blk in os.environ.get("BLOCKS", "1808,1256").split(":"):
    x, y = (int(v) for v in blk.split(","))
    for engine in [int(e) for e in os.environ.get("ENGINES", "2,1").split(",")]:
        abi.set_engine(h, engine)
        r = abi.render_chunk(h, sens, W, H, chunk=(x, y, 8, 8), debug=True)
        st = r["stats"]
        rays = st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
        lines = sorted(set(r["line"].reshape(-1).tolist()))
        print("block %4d,%4d engine %d: kernel %.3f ms, rays %d (secondary %d, shadow %d), wave steps %d, node visits %d (%.1f per ray), box tests %d, tri tests %d, MT tests %d, shaded %d; first-hit lines %s" % (
            x, y, engine, st["kernel_ms"], rays, st["rays_secondary"], st["rays_shadow"], st["wave_node_steps"], st["node_visits"], st["node_visits"] / max(rays, 1),
            st["box_tests"], st["tri_tests"], st["mt_tests"], st["shaded_hits"], lines[:6]), flush=True)
