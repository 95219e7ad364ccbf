"""Work units per frame (MT_DEBUG_PRINT_UNITS=2: no synchronisation between the frames) next to the frame times."""
import os, sys, ctypes, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MT_DEBUG_PRINT_UNITS"] = "2"
import mythtracer_amd as M
sys.path.insert(0, os.path.join(ROOT, "scripts")); import knobs
from mythtracer_amd import scenegen as sg, binding
torch.cuda.init(); torch.zeros(1, device="cuda")
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
abi.set_engine(h, 1); abi.set_stats(h, False)
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
os.environ["MT_DEBUG_BLEND"] = os.environ.get("BLEND", "0"); knobs.from_env(abi, h)
for rep in range(2):
    for i in range(16):
        abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
    torch.cuda.synchronize(); a, b = abi.kernel_times(h)
    print("frames: " + " ".join("%.2f" % x for x in (a + b)), file=sys.stderr, flush=True)
