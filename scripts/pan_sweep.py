"""The bench's moving regime (camera pans in 2-degree steps within +-8 degrees of the golden camera): frame time against
the re-projection radius of the forecast and the cutting thresholds of a re-projected forecast.  Counters off."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
torch.cuda.init(); torch.zeros(1, device="cuda")
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
abi = M.hip_abi(); h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS); abi.set_stats(h, False)
W, H = 1920, 1080
buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
def cam_of(j):
    j %= 16
    tri = j if j <= 4 else (8 - j if j <= 12 else j - 16)
    c = list(sg.ROOM_CAMERA); c[4] += 2.0 * tri
    return c
sens = [binding.sensor(cam_of(j), W, H) for j in range(16)]
def pan(n=48):
    for i in range(8 + n):
        abi.render_chunk_device(h, sens[i % 16], W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
    torch.cuda.synchronize()
    a, b = abi.kernel_times(h)
    t = (a + b)[-n:]
    return float(t.mean()), float(t.max())
abi.set_engine(h, 0)
for i in range(8 + 32):
    abi.render_chunk_device(h, sens[i % 16], W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
torch.cuda.synchronize()
a, b = abi.kernel_times(h)
t = (a + b)[-32:]
print("automatic engine, defaults: pan mean %.3f max %.3f" % (t.mean(), t.max()))
print("per frame (j = position in the 16-frame pan: 0 and 8 = golden camera): " + " ".join("%d:%.2f" % ((8 + k) % 16, t[k]) for k in range(32)))
