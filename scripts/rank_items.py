"""Unit costs of one rank's share (tiles k = rank mod N) after a few frames."""
import ctypes, os, sys, numpy as np, torch, heapq
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
W, H, T = 1920, 1080, 64
world = int(os.environ.get("WORLD", "8")); rank = int(os.environ.get("RANK", "0"))
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
f, s, n = tiling.rank_tiles(W, H, T, T, rank, world)
slots = torch.zeros(max(n, 1) * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda")
for i in range(int(os.environ.get("FRAMES", "5"))):
    abi.render_tiles_device(h, sens, W, H, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
    torch.cuda.synchronize()
    a, b = abi.kernel_times(h)
    arr = np.fromfile(out, dtype=np.uint64).reshape(-1, 2); arr = arr[arr[:, 0] > 0]
    d = arr[:, 0].astype(np.float64); sub = (arr[:, 1] & np.uint64(0xff)).astype(np.int64) - 1
    passes = (arr[:, 1] >> np.uint64(40)).astype(np.int64)
    kinds = np.where(sub < 0, 0, np.where(sub < 4, 1, np.where(sub < 20, 2, 3)))
    hq = [0.0] * 3072; heapq.heapify(hq)
    for x in d: t = heapq.heappop(hq); heapq.heappush(hq, t + x)
    print("frame %d: %.3f+%.3f ms, units %d by kind %s, sum %.3e, max %.3e (kind %d, %d passes), sim makespan %.3e (%.2f ms), balance %.3e" % (
        i, a[-1], b[-1], len(d), np.bincount(kinds, minlength=4).tolist(), d.sum(), d.max(), kinds[d.argmax()], passes[d.argmax()], max(hq), max(hq) / 2.4e6, d.sum() / 3072), flush=True)
    for k in range(4):
        if (kinds == k).any(): print("    kind %d: n %d mean %.3e max %.3e passes mean %.1f" % (k, (kinds == k).sum(), d[kinds == k].mean(), d[kinds == k].max(), passes[kinds == k].mean()))
