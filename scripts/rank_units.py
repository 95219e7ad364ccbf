"""Work units of ONE rank's share of the 4K frame at N = 8 (ray pool): where is the tail?"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "items_rank8.bin")
os.environ["MT_DEBUG_ITEM_CYCLES"] = out
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding, tiling
torch.cuda.init(); torch.zeros(1, device="cuda")
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
W, H, T = 3840, 2160, 64
world = int(os.environ.get("WORLD", "8"))
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
f, s, n = tiling.rank_tiles(W, H, T, T, 0, world)
slots = torch.zeros(max(n, 1) * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda")
for _ in range(5):
    abi.render_tiles_device(h, sens, W, H, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
    torch.cuda.synchronize()
a, b = abi.kernel_times(h)
print("kernel ms (last):", float(b[-1]))
u = np.fromfile(out, dtype=np.uint64).reshape(-1, 2)
u = u[: u.shape[0] // 3]
u = u[u[:, 0] > 0]
cyc = u[:, 0].astype(np.float64); passes = (u[:, 1] >> np.uint64(40)).astype(np.float64); sub = (u[:, 1] & np.uint64(0xff)).astype(np.int64) - 1
print("units %d (whole %d, quarters %d, cells %d); sum %.3e cycles; per wave %.3e; longest %.3e" % (
    len(cyc), (sub < 0).sum(), ((sub >= 0) & (sub < 4)).sum(), (sub >= 4).sum(), cyc.sum(), cyc.sum() / 3072, cyc.max()))
for i in np.argsort(-cyc)[:10]:
    print("  cycles %.3e passes %3d sub %2d  (%.0f cycles per pass)" % (cyc[i], passes[i], sub[i], cyc[i] / passes[i]))
