"""Hysteresis of the cutting decision on a repeated frame (MT_DEBUG_QUAD_KEEP: a block that was rendered in four pieces
stays so while its forecast is above this fraction of the threshold): 64 timed frames per setting, work counters off."""
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
for rep in range(2):
    for keep in sys.argv[1:] or ("1.0", "0.8", "0.6", "0.4"):
        os.environ["MT_DEBUG_QUAD_KEEP"] = keep; knobs.from_env(abi, h)
        abi.set_engine(h, 1)  # (forgets the recorded costs)
        abi.set_stats(h, False)
        for _ in range(32):
            abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize(); abi.kernel_times(h)
        for _ in range(64):
            abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize()
        a, b = abi.kernel_times(h)
        t = a + b
        bad = np.nonzero(t > t.min() * 1.03)[0]
        print("keep %s: mean %.3f median %.3f min %.3f max %.3f; frames above min + 3 %%: %s" % (keep, t.mean(), np.median(t), t.min(), t.max(), list(bad)), flush=True)
