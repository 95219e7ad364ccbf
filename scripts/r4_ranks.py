"""One-GPU estimate of the N-GPU 3840x2160 frame (render only, no exchange): every rank is a scene of its own (its cost
history), frames outside, ranks inside; after every frame the ranks' cost maps are combined (MAX) and imported, as
bench.py does with an all-reduce.  The N-GPU frame time is the slowest rank's.

  OWN=modular|dealt  TILE=64  WORLDS=1,8  REGIME=pan|rest  FRAMES=40  DUMP=path.npy  TUNE=K=V,..  python scripts/r4_ranks.py

OWN=dealt: tiles dealt out by the combined cost map of the previous frame (mt_order_tiles_device, mt_deal_tiles_device,
mt_render_tile_list_device); a camera at rest keeps its lists from the third frame on.  DUMP: the combined cost map
of the last frame (uint32 [270][480]) for offline looks at the assignment."""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding, tiling
torch.cuda.init(); torch.zeros(1, device="cuda")
info = sg.write_scene(os.environ.get("SCENE", "room"), "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
flat = m.flatten()
abi = M.HipAbi(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_%s.so" % os.environ["LIB"])) if os.environ.get("LIB") else M.hip_abi()  # (LIB=<name>: a second build, for A/B runs on one box)
W4, H4, T = 3840, 2160, int(os.environ.get("TILE", "64"))
mw, mh = (W4 + 7) // 8, (H4 + 7) // 8
tx, ty = tiling.tile_grid(W4, H4, T, T)
total = tx * ty
own = os.environ.get("OWN", "dealt")
regime = os.environ.get("REGIME", "pan")
frames = int(os.environ.get("FRAMES", "40"))
vp = lambda t: ctypes.c_void_p(t.data_ptr())
def cam_of(j):
    j %= 16
    tri = j if j <= 4 else (8 - j if j <= 12 else j - 16)
    c = list(sg.ROOM_CAMERA); c[4] += 2.0 * tri if regime == "pan" else 0.0
    return c
s_pan = [binding.sensor(cam_of(j), W4, H4) for j in range(16)]
res = {}
for world in [int(x) for x in os.environ.get("WORLDS", "1,8").split(",")]:
    hs = [abi.scene_create(flat) for _ in range(world)]
    maps = [torch.zeros((mh, mw), dtype=torch.int32, device="cuda") for _ in range(world)]
    comb = torch.zeros((mh, mw), dtype=torch.int32, device="cuda")
    order = torch.zeros(total, dtype=torch.int32, device="cuda")
    for hh in hs:
        abi.set_lights(hh, sg.ROOM_LIGHTS); abi.set_stats(hh, False)
        for kv in [x for x in os.environ.get("TUNE", "").split(",") if x]:
            abi.set_tuning(hh, kv.split("=")[0], float(kv.split("=")[1]))
    n_r = [abi.dealt_tile_count(W4, H4, T, T, world, r) if own == "dealt" else tiling.rank_tiles(W4, H4, T, T, r, world)[2] for r in range(world)]
    lists = [torch.zeros(max(n, 1), dtype=torch.int32, device="cuda") for n in n_r]
    slots = [torch.zeros(max(n, 1) * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda") for n in n_r]
    per = np.zeros((world, frames)); imb = []
    list_id, at_rest = 0, 0
    for i in range(frames):
        sens = s_pan[i % 16]
        at_rest = at_rest + 1 if (i > 0 and s_pan[i % 16] is s_pan[(i - 1) % 16]) or regime == "rest" and i > 0 else 0
        redeal = own == "dealt" and (i == 0 or at_rest < 2 or regime == "pan")
        if redeal:
            list_id += 1
            if i > 0: abi.order_tiles_device(hs[0], vp(comb), mw, mh, W4, H4, T, T, vp(order))
            for r in range(world):
                abi.deal_tiles_device(hs[r], vp(order) if i > 0 else None, W4, H4, T, T, world, r, vp(lists[r]))
        for r in range(world):
            if (world > 1 or own == "dealt") and i > 0:
                abi.import_costs_device(hs[r], vp(comb), mw, mh)
            if own == "dealt":
                abi.render_tile_list_device(hs[r], sens, W4, H4, T, T, vp(lists[r]), n_r[r], list_id, 5, vp(slots[r]))
            else:
                f, st_, n = tiling.rank_tiles(W4, H4, T, T, r, world)
                abi.render_tiles_device(hs[r], sens, W4, H4, T, T, f, st_, n, 5, vp(slots[r]))
            if world > 1 or own == "dealt":
                maps[r].zero_()
                abi.export_costs_device(hs[r], vp(maps[r]), mw, mh)
            torch.cuda.synchronize()
            a, b = abi.kernel_times(hs[r])
            per[r, i] = float(a[-1] + b[-1])
        if world > 1 or own == "dealt":
            comb = torch.stack(maps).max(dim=0).values.contiguous()
            sums = np.array([float(mm.sum()) for mm in maps])
            imb.append(sums.max() / sums.mean())
    for hh in hs:
        abi.scene_destroy(hh)
    skip = min(8, frames // 2)
    if os.environ.get("PERFRAME"):
        print("    per frame (slowest rank: ms): " + " ".join("%d:%.2f" % (int(np.argmax(per[:, i])), per[:, i].max()) for i in range(skip, frames)), flush=True)
    if os.environ.get("MATRIX"):
        for i in range(skip, frames):
            print("    frame %2d (pan step %2d): %s" % (i, i % 16, " ".join("%.2f" % x for x in per[:, i])), flush=True)
    t = float(per[:, skip:].max(axis=0).mean())
    res[world] = t
    print("4K %s camera, %d ranks, tiles %d, ownership %s: frame = slowest rank per frame, mean %.3f ms; ranks' means %s; mean of all %.3f; measured cost sums max/mean %.3f" % (
        regime, world, T, own if world > 1 else "-", t, " ".join("%.2f" % x for x in per[:, skip:].mean(axis=1)), per[:, skip:].mean(),
        float(np.mean(imb[skip:])) if imb else 1.0), flush=True)
    if world > 1 and os.environ.get("DUMP"):
        np.save(os.environ["DUMP"], comb.cpu().numpy().astype(np.uint32))
if 1 in res:
    for world in res:
        if world != 1: print("  -> %d GPUs: %.2fx over one (render only, no exchange)" % (world, res[1] / res[world]), flush=True)
