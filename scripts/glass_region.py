"""Do the blocks with the long ray trees (the glass objects) cost fewer wave cycles in the ray pool than in the state
machine?  Sum of the work units' cycles (MT_DEBUG_ITEM_CYCLES) of a chunk around them and of a plain chunk, either
engine, warm."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "items_glass.bin")
os.environ["MT_DEBUG_ITEM_CYCLES"] = out
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
torch.zeros(1, device="cuda")
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
for name, chunk in (("glass objects", (768, 432, 256, 288)), ("plain wall / floor", (128, 432, 256, 288)), ("whole frame", (0, 0, W, H))):
    for engine in (1, 2):
        abi.set_engine(h, engine)
        for _ in range(10):
            r = abi.render_chunk(h, sens, W, H, chunk=chunk)
        a = np.fromfile(out, dtype=np.uint64).reshape(-1, 2); a = a[a[:, 0] > 0]
        d = a[:, 0].astype(np.float64); passes = (a[:, 1] >> np.uint64(40)).astype(np.float64)
        rays = sum(r["stats"][k] for k in ("rays_primary", "rays_secondary", "rays_shadow"))
        print("%-18s engine %d: units %5d  passes %6.0f  sum %.3e cycles = %.0f per ray  longest %.3e  kernel %.3f ms" % (
            name, engine, len(d), passes.sum(), d.sum(), d.sum() / rays, d.max(), r["stats"]["kernel_ms"]), flush=True)
