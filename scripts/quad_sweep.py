"""Sweep of the state machine's cutting constants on the warm 1080p room frame (MT_DEBUG_QUAD_SHARE / _WORK), work
counters off, 32 frames per setting after 8 settling frames; and the same for the moving camera (2 degrees per frame)."""
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
def frames(cams):
    for cam in cams:
        abi.render_chunk_device(h, binding.sensor(cam, W, H), W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
    torch.cuda.synchronize()
    a, b = abi.kernel_times(h)
    return (a + b)[-len(cams):]
for work in (1.7,) if len(sys.argv) > 1 else (1.5, 1.7, 1.9):
    for share in ([float(x) for x in sys.argv[1:]] or (0.7, 0.8, 0.9)):
        os.environ["MT_DEBUG_QUAD_SHARE"] = str(share); os.environ["MT_DEBUG_QUAD_WORK"] = str(work)
        os.environ["MT_DEBUG_QUAD_SHARE_MOVING"] = str(share - 0.4 if len(sys.argv) > 1 else share)  # (argv: the moving camera 0.4 below)
        knobs.from_env(abi, h)
        abi.set_engine(h, 1); abi.set_stats(h, False)
        frames([sg.ROOM_CAMERA] * 24)
        t = frames([sg.ROOM_CAMERA] * 32)
        mv = frames([sg.ROOM_CAMERA[:4] + (2.0 * i,) + sg.ROOM_CAMERA[5:] for i in range(1, 13)])
        print("quad_work %.1f quad_share %.2f: repeated frame mean %.3f (min %.3f max %.3f) | moving camera mean %.3f max %.3f" % (
            work, share, t.mean(), t.min(), t.max(), mv.mean(), mv.max()), flush=True)
