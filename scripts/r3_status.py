"""The regimes this round's engineering targets, in one run (counters off, every frame checked against the golden):
warm / cold 1080p frame per engine, the slowest of N ranks on the 3840x2160 frame (one-GPU estimate of the N-GPU
frame without the exchange), and the cost of passes with rays that have a zero direction component.

  python scripts/r3_status.py [lib-name ...]     lib/libmythtracer_hip_<name>.so; "now" = the current library
  WHAT=warm,cold,ranks,irr  WORLDS=8  ENGINES=1,2,0
"""
import ctypes, hashlib, json, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding, tiling
torch.cuda.init(); torch.zeros(1, device="cuda")
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
flat = m.flatten()
golden = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["room_1920x1080_d5"]["sha256"]
what = os.environ.get("WHAT", "warm,cold,ranks,irr").split(",")
engines = [int(x) for x in os.environ.get("ENGINES", "1,2,0").split(",")]
ENG = {0: "automatic", 1: "state machine", 2: "ray pool", 3: "hybrid"}
for name in (sys.argv[1:] or ["now"]):
    abi = M.hip_abi() if name == "now" else M.HipAbi(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_%s.so" % name))
    h = abi.scene_create(flat); abi.set_lights(h, sg.ROOM_LIGHTS); abi.set_stats(h, False)
    W, H = 1920, 1080
    sens = binding.sensor(sg.ROOM_CAMERA, W, H)
    buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    def frame(s12=sens):
        abi.render_chunk_device(h, s12, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize()
        a, b = abi.kernel_times(h)
        return float(a[-1] + b[-1])
    def ok():
        return hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest() == golden
    for e in engines:
        if "warm" in what or "cold" in what:
            abi.set_engine(h, e)
            cold = []
            for _ in range(3):
                abi.set_scheduling(h, True)
                cold.append(frame())
            c_ok = ok()
            t = [frame() for _ in range(40)]
            print("[%s] %-13s 1080p: cold %s ms (%s) | warm mean(last 16) %.3f min %.3f (%s)" % (
                name, ENG[e], " ".join("%.2f" % x for x in cold), "ok" if c_ok else "MISMATCH",
                sum(t[-16:]) / 16, min(t), "ok" if ok() else "MISMATCH"), flush=True)
            if e == 0:
                mv = []
                for f in range(1, 25):
                    c = list(sg.ROOM_CAMERA); c[4] += 2.0 * (f if f <= 8 else 16 - f if f <= 16 else f - 24)
                    mv.append(frame(binding.sensor(c, W, H)))
                print("[%s] %-13s 1080p: camera panning 2 deg per frame within +-16 deg: mean %.3f max %.3f" % (name, ENG[e], sum(mv) / len(mv), max(mv)), flush=True)
    if "ranks" in what:
        W4, H4, T = 3840, 2160, int(os.environ.get("TILE", "64"))
        s4 = binding.sensor(sg.ROOM_CAMERA, W4, H4)
        for world in [int(x) for x in os.environ.get("WORLDS", "8").split(",")]:
            for e in engines:
                abi.set_engine(h, e)
                per = []
                for rank in range(world):
                    f, s, n = tiling.rank_tiles(W4, H4, T, T, rank, world)
                    slots = torch.zeros(max(n, 1) * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda")
                    for _ in range(4):
                        abi.render_tiles_device(h, s4, W4, H4, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
                    torch.cuda.synchronize(); abi.kernel_times(h)
                    for _ in range(4):
                        abi.render_tiles_device(h, s4, W4, H4, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
                    torch.cuda.synchronize()
                    a, b = abi.kernel_times(h)
                    per.append(float((a + b).mean()))
                print("[%s] %-13s 4K, %d ranks: slowest %.3f ms (rank %d), fastest %.3f, mean %.3f" % (
                    name, ENG[e], world, max(per), int(np.argmax(per)), min(per), sum(per) / len(per)), flush=True)
    if "pan" in what:
        # the bench's regime at N ranks: the camera pans 2 degrees per frame within +-8 degrees; every rank is a scene
        # of its own here (its cost history), frames outside, ranks inside; after every frame the ranks' cost maps are
        # combined (MAX) and imported, as bench.py does with an all-reduce.  The N-GPU frame time is the slowest rank's.
        W4, H4, T = 3840, 2160, int(os.environ.get("TILE", "64"))
        mw, mh = (W4 + 7) // 8, (H4 + 7) // 8
        def cam_of(j):
            j %= 16
            tri = j if j <= 4 else (8 - j if j <= 12 else j - 16)
            c = list(sg.ROOM_CAMERA); c[4] += 2.0 * tri
            return c
        s_pan = [binding.sensor(cam_of(j), W4, H4) for j in range(16)]
        res = {}
        for exchange in ((True, False) if os.environ.get("NOEXCHANGE") else (True,)):
            for world in [int(x) for x in os.environ.get("WORLDS", "1,8").split(",")]:
                hs = [abi.scene_create(flat) for _ in range(world)]
                maps = [torch.zeros((mh, mw), dtype=torch.int32, device="cuda") for _ in range(world)]
                comb = torch.zeros((mh, mw), dtype=torch.int32, device="cuda")
                for hh in hs:
                    abi.set_lights(hh, sg.ROOM_LIGHTS); abi.set_stats(hh, False)
                    for kv in [x for x in os.environ.get("TUNE", "").split(",") if x]:
                        abi.set_tuning(hh, kv.split("=")[0], float(kv.split("=")[1]))
                geo = [tiling.rank_tiles(W4, H4, T, T, r, world) for r in range(world)]
                slots = [torch.zeros(max(g[2], 1) * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda") for g in geo]
                per = np.zeros((world, 40))
                for i in range(40):
                    for r in range(world):
                        f, st_, n = geo[r]
                        if exchange and world > 1 and i > 0:
                            abi.import_costs_device(hs[r], ctypes.c_void_p(comb.data_ptr()), mw, mh)
                        abi.render_tiles_device(hs[r], s_pan[i % 16], W4, H4, T, T, f, st_, n, 5, ctypes.c_void_p(slots[r].data_ptr()))
                        if exchange and world > 1:
                            maps[r].zero_()
                            abi.export_costs_device(hs[r], ctypes.c_void_p(maps[r].data_ptr()), mw, mh)
                        torch.cuda.synchronize()
                        a, b = abi.kernel_times(hs[r])
                        per[r, i] = float(a[-1] + b[-1])
                    if exchange and world > 1:
                        comb = torch.stack(maps).max(dim=0).values.contiguous()
                for hh in hs:
                    abi.scene_destroy(hh)
                if os.environ.get("PERFRAME"):
                    print("    per frame (slowest rank: ms): " + " ".join("%d:%.1f" % (int(np.argmax(per[:, i])), per[:, i].max()) for i in range(8, 40)), flush=True)
                t = float(per[:, 8:].max(axis=0).mean())
                if exchange: res[world] = t
                print("[%s] automatic 4K panning camera, %d ranks%s: frame = slowest rank per frame, mean %.3f ms (ranks' means %s)" % (
                    name, world, "" if exchange else " WITHOUT the cost-map exchange", t, " ".join("%.2f" % x for x in per[:, 8:].mean(axis=1))), flush=True)
        if 1 in res:
            for world in res:
                if world != 1: print("[%s]   -> %d GPUs: %.2fx over one (render only, no exchange)" % (name, world, res[1] / res[world]), flush=True)
    if "irr" in what:
        rnd = np.random.RandomState(1)
        n_waves = 2048
        n = 64 * n_waves
        for per_wave in (1, 4, 8, 16, 64):
            rays = np.zeros((n, 6))
            rays[:, :3] = 1.0e6; rays[:, 3:] = (0.6, 0.64, 0.48)  # away from the scene
            for w in range(n_waves):
                ang = rnd.uniform(-0.6, 0.6, per_wave)
                k = w * 64 + np.arange(per_wave) * (64 // per_wave)
                rays[k, 0] = 200.0; rays[k, 1] = 120.0; rays[k, 2] = 20.0
                d = np.stack([np.zeros(per_wave), np.sin(ang) * 0.5, np.cos(ang)], axis=1)
                d /= np.linalg.norm(d, axis=1)[:, None]; d[:, 0] = 0.0
                rays[k, 3:] = d
            t = min(abi.intersect_rays(h, rays)["stats"]["kernel_ms"] for _ in range(3))
            print("[%s] %2d rays with a zero direction component per wave, %d waves: %.3f ms = %.2f M cycles per wave" % (
                name, per_wave, n_waves, t, t * 2.4), flush=True)
    abi.scene_destroy(h)
