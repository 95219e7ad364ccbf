"""What ONE rank of N renders of the 3840x2160 frame (tiles k = rank mod N), timed on this GPU
with either engine: the N-GPU frame time without the gather."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding, tiling
torch.cuda.init(); torch.zeros(1, device="cuda")
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
W, H, T = 3840, 2160, 64
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
for world in [int(x) for x in os.environ.get("WORLDS", "1,2,4,8").split(",")]:
    for engine in [int(x) for x in os.environ.get("ENGINES", "1,2").split(",")]:
        abi.set_engine(h, engine)
        worst, who = 0.0, -1
        for rank in range(world):  # every rank: the ones whose tiles hold the pixel column x = W / 2 (rays with a zero direction component) are the slowest
            f, s, n = tiling.rank_tiles(W, H, T, T, rank, world)
            slots = torch.zeros(max(n, 1) * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda")
            for _ in range(4):
                abi.render_tiles_device(h, sens, W, H, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
            torch.cuda.synchronize(); abi.kernel_times(h)
            for _ in range(4):
                abi.render_tiles_device(h, sens, W, H, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
            torch.cuda.synchronize()
            a, b = abi.kernel_times(h)
            if float((a + b).mean()) > worst: worst, who = float((a + b).mean()), rank
        print("4K, world %d, engine %s: slowest rank (%d) %.3f ms per frame" % (world, "state machine" if engine == 1 else "ray pool", who, worst), flush=True)
