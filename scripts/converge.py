"""Frame times from the first frame of a geometry on (no measured costs at first): how fast does the schedule settle?"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
torch.cuda.init(); torch.zeros(1, device="cuda")
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); abi = M.hip_abi(); flat = m.flatten()
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
for rep in range(3):
    h = abi.scene_create(flat); abi.set_lights(h, sg.ROOM_LIGHTS); abi.set_stats(h, False)
    for _ in range(16):
        abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
    torch.cuda.synchronize(); a, b = abi.kernel_times(h); t = a + b
    print("frames 1..16 (counters off): " + " ".join("%.2f" % x for x in t[-16:]), flush=True)
    abi.scene_destroy(h)
