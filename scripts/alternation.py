import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, "/root/repo")
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
def run(tag, n=16):
    for i in range(n):
        abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
    torch.cuda.synchronize(); a, b = abi.kernel_times(h)
    print(tag + ": " + " ".join("%.2f" % x for x in (a + b)), flush=True)
run("warm-up"); run("history on")
abi.set_scheduling(h, False); run("history off (classified)"); run("history off (classified)")
abi.set_scheduling(h, True); run("history on again"); 
os.environ["MT_DEBUG_BLEND"] = "1.0"; knobs.from_env(abi, h)   # forecast frozen after the first two frames
run("forecast frozen (blend 1)"); run("forecast frozen (blend 1)")
