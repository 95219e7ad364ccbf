"""Sweep of the hybrid engine's thresholds (engine 3): blocks above HYBRID_POOL_SHARE of an even split go to the ray
pool in pieces, above HYBRID_QUAD_SHARE to the state machine as quarters.  Warm 1080p frame and the slowest of 8
ranks on the 4K frame, counters off."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding, tiling
torch.cuda.init(); torch.zeros(1, device="cuda")
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
abi = M.hip_abi(); h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS); abi.set_stats(h, False)
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
W4, H4, T = 3840, 2160, 64
s4 = binding.sensor(sg.ROOM_CAMERA, W4, H4)
slots = torch.zeros(260 * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda")
def warm():
    for _ in range(28):
        abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
    torch.cuda.synchronize()
    a, b = abi.kernel_times(h)
    t = (a + b)[-12:]
    return float(t.mean()), float(t.max())
def ranks(world=8, which=None):
    per = []
    for rank in (which or range(world)):
        f, s, n = tiling.rank_tiles(W4, H4, T, T, rank, world)
        for _ in range(7):
            abi.render_tiles_device(h, s4, W4, H4, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
        torch.cuda.synchronize()
        a, b = abi.kernel_times(h)
        per.append(float((a + b)[-3:].mean()))
    return per
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
def full4k():
    b4 = torch.zeros((H4, W4, 3), dtype=torch.uint8, device="cuda")
    for _ in range(10):
        abi.render_chunk_device(h, s4, W4, H4, (0, 0, W4, H4), 5, ctypes.c_void_p(b4.data_ptr()))
    torch.cuda.synchronize()
    a, b = abi.kernel_times(h)
    return float((a + b)[-4:].mean())
if mode == "cells":
    for cf in (3.0, 1.5, 1.0, 0.6, 0.35):
        abi.set_tuning(h, "HYBRID_CELL_FACTOR", cf)
        for pt2 in (0.12, 0.2):
            abi.set_tuning(h, "POOL_PIECE_TIME2", pt2)
            for ps in (0.9, 1.0):
                abi.set_engine(h, 3)
                abi.set_tuning(h, "HYBRID_POOL_SHARE", ps); abi.set_tuning(h, "HYBRID_QUAD_SHARE", ps)
                per = ranks()
                print("cell_factor %.2f piece_time2 %.2f pool_share %.2f: 4K 8 ranks slowest %.3f (rank %d) mean %.3f" % (cf, pt2, ps, max(per), int(np.argmax(per)), sum(per) / len(per)), flush=True)
    sys.exit(0)
if mode == "fine":
    grid = [(3, [(ps, qs, w1) for ps in (0.7, 0.8, 0.9, 1.0, 1.15, 1.3) for qs in (ps, 0.95) for w1 in (1.3,)] + [(1.0, 1.0, 1.0), (1.0, 1.0, 1.8), (1.15, 1.15, 1.0), (1.15, 1.15, 1.8)])]
    mode = "ranks"
    for e in (1, 3):
        abi.set_engine(h, e)
        print("engine %d: the whole 4K frame on this GPU %.3f ms" % (e, full4k()), flush=True)
else:
    grid = ((1, [(0, 0, 0)]), (2, [(0, 0, 0)]), (3, [(ps, qs, w1) for ps in (0.6, 0.9, 1.3, 2.0, 3.0) for qs in (0.6, 0.95) for w1 in (1.3,)]))
for engine, settings in grid:
    for ps, qs, w1 in settings:
        abi.set_engine(h, engine)
        if engine == 3:
            abi.set_tuning(h, "HYBRID_POOL_SHARE", ps); abi.set_tuning(h, "HYBRID_QUAD_SHARE", min(qs, ps)); abi.set_tuning(h, "HYBRID_WORK1", w1)
        line = "engine %d pool_share %.2f quad_share %.2f work1 %.1f:" % (engine, ps, min(qs, ps), w1)
        if mode in ("both", "warm"):
            line += " warm 1080p mean %.3f max %.3f |" % warm()
        if mode in ("both", "ranks"):
            per = ranks()
            line += " 4K 8 ranks slowest %.3f (rank %d) mean %.3f" % (max(per), int(np.argmax(per)), sum(per) / len(per))
        print(line, flush=True)
