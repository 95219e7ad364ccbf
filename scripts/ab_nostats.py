"""A/B of library variants (lib/libmythtracer_hip_<name>.so) against the current one on the repeated 1080p room frame,
work counters OFF (the kernels bench.py times), 40 frames each after 8 settling frames."""
import ctypes, hashlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
torch.cuda.init(); torch.zeros(1, device="cuda")
names = sys.argv[1:] + ["now"]
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
flat = m.flatten()
buf = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
for rep in range(2):
    for name in names:
        path = None if name == "now" else os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_%s.so" % name)
        abi = M.HipAbi(path) if path else M.hip_abi()
        h = abi.scene_create(flat); abi.set_lights(h, sg.ROOM_LIGHTS)
        for engine in (1, 2):
            abi.set_engine(h, engine); abi.set_stats(h, False)
            for _ in range(8):
                abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
            torch.cuda.synchronize(); abi.kernel_times(h)
            for _ in range(40):
                abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
            torch.cuda.synchronize(); a, b = abi.kernel_times(h); t = a + b
            sha = hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:12]
            print("%-5s engine %d: mean %.3f median %.3f min %.3f max %.3f ms  frame %s" % (name, engine, t.mean(), np.median(t), t.min(), t.max(), sha), flush=True)
        abi.lib.mt_scene_destroy(h)
