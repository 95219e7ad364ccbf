"""Damping of the cost forecast of a repeated frame (MT_DEBUG_BLEND): 64 frames without work counters per setting."""
import os, sys, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
sys.path.insert(0, os.path.join(ROOT, "scripts")); import knobs
from mythtracer_amd import scenegen as sg, binding
torch.cuda.init(); torch.zeros(1, device="cuda")
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
for engine in (1,):
    for blend in (0.8, 0.9, 0.95, 0.98, 1.0):
        os.environ["MT_DEBUG_BLEND"] = str(blend); knobs.from_env(abi, h)
        abi.set_engine(h, engine)  # (forgets the recorded costs)
        abi.set_stats(h, False)
        for _ in range(12):
            abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize(); abi.kernel_times(h)
        for _ in range(96):
            abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize()
        a, b = abi.kernel_times(h)
        t = a + b
        print("engine %d blend %.2f: mean %.3f min %.3f max %.3f, frames above min + 3 %%: %d of %d" % (engine, blend, t.mean(), t.min(), t.max(), (t > t.min() * 1.03).sum(), len(t)), flush=True)
