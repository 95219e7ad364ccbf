"""What one rank of N renders (tiles k = rank mod N of the 1080p frame), timed on
this GPU: an estimate of the N-GPU frame time without the gather."""
import ctypes, os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding, tiling
torch.cuda.init(); torch.zeros(1, device="cuda")
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
W, H, T = 1920, 1080, int(os.environ.get("TILE", "64"))
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in range(world):
        f, s, n = tiling.rank_tiles(W, H, T, T, rank, world)
        slots = torch.zeros(max(n, 1) * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda")
        for _ in range(4):
            abi.render_tiles_device(h, sens, W, H, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
        torch.cuda.synchronize(); abi.kernel_times(h)
        for _ in range(4):
            abi.render_tiles_device(h, sens, W, H, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
        torch.cuda.synchronize()
        a, b = abi.kernel_times(h)
        worst = max(worst, float((a + b).mean()))
    print("world %d tile %d: slowest rank %.3f ms per frame" % (world, T, worst), flush=True)
