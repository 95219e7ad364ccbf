"""A block near the cutting threshold changes its form every few frames of a repeated frame; with the block's own ratio of
its two measured costs (forecast_item in mt_order.h, item_whole / item_qsum) it does not.  MT_DEBUG_NO_FORMS=1 switches that off.
64 timed frames per setting after 32, work counters off; then the moving camera (yaw += 2 degrees per frame)."""
import os, sys, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
sys.path.insert(0, os.path.join(ROOT, "scripts")); import knobs
from mythtracer_amd import scenegen as sg, binding
torch.cuda.init(); torch.zeros(1, device="cuda")
W, H = 1920, 1080
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
def run(cams, settle, timed):
    abi.set_engine(h, 1); abi.set_stats(h, False)
    for k in range(settle + timed):
        if k == settle:
            torch.cuda.synchronize(); abi.kernel_times(h)
        sens = binding.sensor(cams(k), W, H)
        abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
    torch.cuda.synchronize()
    a, b = abi.kernel_times(h)
    return a + b
for rep in range(2):
    for off in ("1", ""):
        if off: os.environ["MT_DEBUG_NO_FORMS"] = "1"
        else: os.environ.pop("MT_DEBUG_NO_FORMS", None)
        knobs.from_env(abi, h)
        t = run(lambda k: sg.ROOM_CAMERA, 32, 64)
        bad = np.nonzero(t > t.min() * 1.03)[0]
        print("ratio %-3s at rest: mean %.3f median %.3f min %.3f max %.3f; frames above min + 3 %%: %d" % ("off" if off else "on", t.mean(), np.median(t), t.min(), t.max(), len(bad)), flush=True)
        t = run(lambda k: sg.ROOM_CAMERA[:4] + (2.0 * k,) + sg.ROOM_CAMERA[5:], 4, 24)
        print("ratio %-3s moving : mean %.3f median %.3f min %.3f max %.3f" % ("off" if off else "on", t.mean(), np.median(t), t.min(), t.max()), flush=True)
