"""(STALE since round 3: the per-unit dump has the layout scripts/unit_timeline.py decodes; this decoder prints garbage.)
The longest work units of one rank's share of the 3840x2160 frame (tiles k = rank mod N, ray pool by default):
where they are in the picture, how many passes they take."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "items_rank.bin")
os.environ["MT_DEBUG_ITEM_CYCLES"] = out
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding, tiling
torch.zeros(1, device="cuda")
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
W, H, T = 3840, 2160, 64
world = int(os.environ.get("WORLD", "8")); rank = int(os.environ.get("RANK", "0"))
abi.set_engine(h, int(os.environ.get("ENGINE", "2")))  # (the per-unit dump is the ray pool's; the hybrid kernel's stamps have another layout)
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
f, s, n = tiling.rank_tiles(W, H, T, T, rank, world)
tiles_x = (W + T - 1) // T
slots = torch.zeros(max(n, 1) * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda")
for i in range(int(os.environ.get("FRAMES", "6"))):
    abi.render_tiles_device(h, sens, W, H, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
torch.cuda.synchronize()
a, b = abi.kernel_times(h)
arr = np.fromfile(out, dtype=np.uint64).reshape(-1, 2); arr = arr[arr[:, 0] > 0]
d = arr[:, 0].astype(np.float64); sub = (arr[:, 1] & np.uint64(0xff)).astype(np.int64) - 1
item = ((arr[:, 1] >> np.uint64(8)) & np.uint64(0xffffffff)).astype(np.int64)
passes = (arr[:, 1] >> np.uint64(40)).astype(np.int64)
per_tile = (T // 8) * (T // 8)
tile = f + (item // per_tile) * s
bx = (tile % tiles_x) * T + (item % per_tile % (T // 8)) * 8; by = (tile // tiles_x) * T + (item % per_tile // (T // 8)) * 8
print("last frame %.3f + %.3f ms; units %d, sum %.3e cycles, even share of 2048 waves %.3e, longest %.3e" % (a[-1], b[-1], len(d), d.sum(), d.sum() / 2048, d.max()))
for k in np.argsort(-d)[:14]:
    print("  %.3e cycles  %2d passes  sub %2d  block x %4d y %4d%s" % (d[k], passes[k], sub[k], bx[k], by[k], "   <- pixel column x = %d" % (W // 2) if bx[k] <= W // 2 < bx[k] + 8 else ""))
