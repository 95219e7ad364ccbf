"""rank_share4k for a given library (LIB=path) and engine."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding, tiling
torch.cuda.init(); torch.zeros(1, device="cuda")
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); flat = m.flatten()
W, H, T = 3840, 2160, 64
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
for lib in os.environ["LIBS"].split(","):
    abi = M.HipAbi(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_%s.so" % lib))
    h = abi.scene_create(flat); abi.set_lights(h, sg.ROOM_LIGHTS)
    for world in [int(x) for x in os.environ.get("WORLDS", "8").split(",")]:
        for engine in (1, 2):
            abi.set_engine(h, engine)
            f, s, n = tiling.rank_tiles(W, H, T, T, 0, world)
            slots = torch.zeros(max(n, 1) * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda")
            for _ in range(5):
                abi.render_tiles_device(h, sens, W, H, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
            torch.cuda.synchronize(); abi.kernel_times(h)
            for _ in range(4):
                abi.render_tiles_device(h, sens, W, H, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
            torch.cuda.synchronize()
            a, b = abi.kernel_times(h)
            print("lib %s world %d engine %d: %.3f ms" % (lib, world, engine, float((a + b).mean())), flush=True)
