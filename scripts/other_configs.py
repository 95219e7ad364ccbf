"""Frame times of the other BASELINE configurations on one GPU (configs[1]: 1280x720 primary rays only)."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
abi = M.hip_abi(); h = m.device_scene()
for name, W, H, lights, depth in (("configs[1] 1280x720, primary rays only", 1280, 720, [], 0),
                                  ("configs[1] 1280x720, 1 light", 1280, 720, sg.ROOM_LIGHTS[:1], 0),
                                  ("configs[3] 1920x1080, 3 lights, depth 4", 1920, 1080, sg.ROOM_LIGHTS, 4),
                                  ("configs[4] frame 3840x2160 on one GPU", 3840, 2160, sg.ROOM_LIGHTS, 5)):
    abi.set_lights(h, lights)
    sens = binding.sensor(sg.ROOM_CAMERA, W, H)
    rs = [abi.render_chunk(h, sens, W, H, max_depth=depth)["stats"] for _ in range(7)]
    t = [r["kernel_ms"] for r in rs]
    rays = sum(rs[-1][k] for k in ("rays_primary", "rays_secondary", "rays_shadow"))
    print("%s: cold %.2f ms, warm median %.2f ms (work counters on), %d rays -> %.0f Mray/s" % (name, t[0], float(np.median(t[2:])), rays, rays / float(np.median(t[2:])) / 1e3), flush=True)
