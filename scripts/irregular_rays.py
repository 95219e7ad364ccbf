"""Cost of rays with one zero direction component (OctTree::IntersectRay batches): in the plane x = 200 of the room
(the coordinate the host's maps cover after a render from the default camera), in a plane the maps do not cover, and
the same rays tilted out of the plane by 1e-12 (regular: hit-set walk)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
abi.render_chunk(h, binding.sensor(sg.ROOM_CAMERA, 64, 36), 64, 36)  # the maps for the camera's origin
rnd = np.random.RandomState(1)
n = 64 * 512
ang = rnd.uniform(-0.6, 0.6, n)
for name, x, dx in (("in the plane x = 200 (maps)", 200.0, 0.0), ("in the plane x = 200.25 (no maps)", 200.25, 0.0), ("tilted by 1e-12 (regular)", 200.0, 1e-12)):
    rays = np.zeros((n, 6))
    rays[:, 0] = x; rays[:, 1] = 120.0; rays[:, 2] = 20.0
    rays[:, 3] = dx; rays[:, 4] = np.sin(ang) * 0.5; rays[:, 5] = np.cos(ang)
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1)[:, None]
    rays[:, 3] = dx  # (keep the exact zero)
    if name.startswith("in the plane x = 200.25"):
        t0 = min(abi.intersect_rays(h, rays)["stats"]["kernel_ms"] for _ in range(3))
        abi.render_chunk(h, binding.sensor((x,) + tuple(sg.ROOM_CAMERA[1:]), 64, 36), 64, 36)  # maps for THIS coordinate now
        name = "in the plane x = 200.25 (%.3f ms without maps; with)" % t0
    t = []
    for _ in range(3):
        r = abi.intersect_rays(h, rays)
        t.append(r["stats"]["kernel_ms"])
    st = r["stats"]
    print("%-58s: %.3f ms for %d rays; hits %.2f; node visits / ray %.1f, triangle tests / ray %.0f, Moeller-Trumbore / ray %.2f" % (
        name, min(t), n, (r["line"] >= 0).mean(), st["node_visits"] / n, st["tri_tests"] / n, st["mt_tests"] / n), flush=True)
