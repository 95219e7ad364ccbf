"""Cutting threshold of a RE-PROJECTED forecast (camera turning 2 degrees per frame; MT_DEBUG_QUAD_SHARE_MOVING) and the
work factor of its quartered blocks (MT_DEBUG_QUAD_WORK): 36 frames per setting, work counters off."""
import ctypes, os, sys, numpy as np, torch
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
    out = []
    for cam in cams:
        abi.render_chunk_device(h, binding.sensor(cam, W, H), W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize(); a, b = abi.kernel_times(h); out.append(float(a[-1] + b[-1]))
    return np.array(out)
for rep in range(2):
    for work in ("1.5",):
        for share in sys.argv[1:] or ("0.55", "0.6", "0.65", "0.7", "0.75", "0.8"):
            os.environ["MT_DEBUG_QUAD_SHARE_MOVING"] = share; os.environ["MT_DEBUG_QUAD_WORK"] = work; knobs.from_env(abi, h)
            abi.set_engine(h, 1); abi.set_stats(h, False)
            frames([sg.ROOM_CAMERA] * 6)
            mv = frames([sg.ROOM_CAMERA[:4] + (2.0 * i,) + sg.ROOM_CAMERA[5:] for i in range(1, 37)])
            print("moving share %s work %s: mean %.3f median %.3f max %.3f" % (share, work, mv.mean(), np.median(mv), mv.max()), flush=True)
