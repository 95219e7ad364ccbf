"""Where a pass with FOUR rays that have a zero direction component spends its cycles (diagnostic -DMT_PROF library):
waves of 4 such rays in the plane x = 200 of the room + 60 rays that miss the scene."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
lib = sys.argv[1] if len(sys.argv) > 1 else "prof"
abi = M.hip_abi() if lib == "now" else M.HipAbi(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_%s.so" % lib))
h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS)
rnd = np.random.RandomState(1)
n_waves = 2048
n = 64 * n_waves
for per_wave in (4, 1, 16, 64):
    rays = np.zeros((n, 6))
    rays[:, 0] = 1.0e6; rays[:, 1] = 1.0e6; rays[:, 2] = 1.0e6; rays[:, 3:] = (0.6, 0.64, 0.48)  # away from the scene
    for w in range(n_waves):
        ang = rnd.uniform(-0.6, 0.6, per_wave)
        k = w * 64 + np.arange(per_wave) * (64 // per_wave)
        rays[k, 0] = 200.0; rays[k, 1] = 120.0; rays[k, 2] = 20.0
        d = np.stack([np.zeros(per_wave), np.sin(ang) * 0.5, np.cos(ang)], axis=1)
        d /= np.linalg.norm(d, axis=1)[:, None]; d[:, 0] = 0.0
        rays[k, 3:] = d
    t = min(abi.intersect_rays(h, rays)["stats"]["kernel_ms"] for _ in range(3))
    print("%2d such rays per wave, %d waves: %.3f ms = %.2f M cycles per wave at 2.4 GHz (one wave per SIMD slot pair: %d waves at once)" % (
        per_wave, n_waves, t, t * 2.4e6 / 1e6, 2048), flush=True)
