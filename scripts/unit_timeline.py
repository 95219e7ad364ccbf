"""When do the work units of a launch start and end?  (-DMT_DEBUG_KNOBS build: MT_DEBUG_ITEM_CYCLES dumps per unit its
duration, and its start stamp and wave.)  Prints the launch's makespan in s_memtime ticks, the units that end last, the
waves' idle time at the end, and how much of the launch had fewer than all waves busy.

  WHAT=1080p|rank  RANK=5 WORLD=8  ENGINE=3  YAW=0  COLD=  python scripts/unit_timeline.py"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "items_tl.bin")
os.environ["MT_DEBUG_ITEM_CYCLES"] = out
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding, tiling
torch.zeros(1, device="cuda")
info = sg.write_scene(os.environ.get("SCENE", "room"), "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
abi = M.HipAbi(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_knobs.so")); h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS); abi.set_stats(h, False)
abi.set_engine(h, int(os.environ.get("ENGINE", "3")))
if os.environ.get("TRAV"): abi.set_traversal_mode(h, int(os.environ["TRAV"]))  # 3 = the ordered descent for every ray
for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]:
    abi.set_tuning(h, k, float(v))
what = os.environ.get("WHAT", "rank")
if what == "1080p":
    W, H = 1920, 1080
    cam1080 = list(sg.ROOM_CAMERA); cam1080[4] += float(os.environ.get("YAW", "0"))  # (YAW=2: no pixel column of zero-component rays)
    sens = binding.sensor(cam1080, W, H)
    buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    n_items = (W // 8) * (H // 8)
    for i in range(1 if os.environ.get("COLD") else 10):  # (COLD=1: the first frame of a geometry -- probe_kernel's guess, ray pool)
        abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
else:
    W, H, T = 3840, 2160, int(os.environ.get("TILE", "64"))
    world = int(os.environ.get("WORLD", "8")); rank = int(os.environ.get("RANK", "5"))
    sens = binding.sensor(sg.ROOM_CAMERA, W, H)
    f, s, n = tiling.rank_tiles(W, H, T, T, rank, world)
    n_items = n * (T // 8) * (T // 8)
    slots = torch.zeros(max(n, 1) * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda")
    if os.environ.get("PAN"):
        # the bench's regime: the camera pans 2 degrees per frame, all ranks exchange their cost maps after every frame
        # (the other ranks through the shipping library, this one through the knobs build); the LAST frame is dumped --
        # PAN=<number of frames>, e.g. 21 ends 2 degrees off the golden camera, 25 on it (the zero-component column)
        abi0 = M.hip_abi(); flat = m.flatten()
        mw, mh = (W + 7) // 8, (H + 7) // 8
        hs = [h if r == rank else abi0.scene_create(flat) for r in range(world)]
        for r in range(world):
            if r != rank: abi0.set_lights(hs[r], sg.ROOM_LIGHTS); abi0.set_stats(hs[r], False)
        geo = [tiling.rank_tiles(W, H, T, T, r, world) for r in range(world)]
        sl = [torch.zeros(max(g[2], 1) * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda") for g in geo]
        maps = [torch.zeros((mh, mw), dtype=torch.int32, device="cuda") for _ in range(world)]
        comb = None
        def cam_of(j):
            j %= 16
            tri = j if j <= 4 else (8 - j if j <= 12 else j - 16)
            c = list(sg.ROOM_CAMERA); c[4] += 2.0 * tri
            return c
        for i in range(int(os.environ["PAN"])):
            sens = binding.sensor(cam_of(i), W, H)
            for r in range(world):
                A = abi if r == rank else abi0
                if comb is not None: A.import_costs_device(hs[r], ctypes.c_void_p(comb.data_ptr()), mw, mh)
                A.render_tiles_device(hs[r], sens, W, H, T, T, geo[r][0], geo[r][1], geo[r][2], 5, ctypes.c_void_p(sl[r].data_ptr()))
                maps[r].zero_(); A.export_costs_device(hs[r], ctypes.c_void_p(maps[r].data_ptr()), mw, mh)
                torch.cuda.synchronize()
            comb = torch.stack(maps).max(dim=0).values.contiguous()
    else:
        for i in range(8):
            abi.render_tiles_device(h, sens, W, H, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
torch.cuda.synchronize()
a, b = abi.kernel_times(h)
raw = np.fromfile(out, dtype=np.uint64)
dur = raw[:n_items * 32].reshape(-1, 2)
st = raw[n_items * 96: n_items * 96 + n_items * 32].reshape(-1, 2)
ok = dur[:, 0] > 0
d = dur[ok, 0].astype(np.float64); t0 = st[ok, 0].astype(np.float64); wave = st[ok, 1].astype(np.int64)
meta = dur[ok, 1]
sub = (meta & np.uint64(0xff)).astype(np.int64) - 1
item = ((meta >> np.uint64(8)) & np.uint64(0xffffffff)).astype(np.int64)
if what == "1080p":
    bx = (item % (W // 8)) * 8; by = (item // (W // 8)) * 8
else:
    per_tile = (T // 8) * (T // 8); tiles_x = (W + T - 1) // T
    tile = f + (item // per_tile) * s
    bx = (tile % tiles_x) * T + (item % per_tile % (T // 8)) * 8; by = (tile // tiles_x) * T + (item % per_tile // (T // 8)) * 8
passes = (meta >> np.uint64(40)).astype(np.int64)
# s_memtime counts per XCD (eight unsynchronised counters, and which workgroup runs on which XCD is the dispatcher's
# business): every WAVE's stamps are taken relative to its own first unit -- the launch fills the chip with persistent
# waves that all start within the dispatch ramp (a few microseconds), so a wave's first unit starts at the launch's start
for wv in np.unique(wave):
    sel = wave == wv
    t0[sel] -= t0[sel].min()
t1 = t0 + d
span = t1.max()
print("%s engine %s: last launch %.3f + %.3f ms; %d units on %d waves; makespan %.3e ticks, sum of units %.3e = %.3f of makespan x 2048 waves" % (
    what, os.environ.get("ENGINE", "3"), a[-1], b[-1], len(d), len(np.unique(wave)), span, d.sum(), d.sum() / (span * 2048)))
ticks_per_ms = span / b[-1]
print("ticks per ms of the frame kernel: %.0f; busy share of the waves %.3f" % (ticks_per_ms, d.sum() / (span * len(np.unique(wave)))))
for lo, hi in ((0.0, 0.5), (0.5, 0.7), (0.7, 0.8), (0.8, 0.9), (0.9, 1.0)):
    a_, b_ = lo * span, hi * span
    busy = (np.minimum(t1, b_) - np.maximum(t0, a_)).clip(min=0).sum() / ((b_ - a_) * len(np.unique(wave)))
    print("  between %.2f and %.2f of the makespan %.3f of the waves are busy" % (lo, hi, busy))
wave_end = np.zeros(wave.max() + 1); np.maximum.at(wave_end, wave, t1)
wave_end = wave_end[wave_end > 0]
print("waves: last unit ends at mean %.3f of the makespan, median %.3f, 10th percentile %.3f" % (wave_end.mean() / span, np.median(wave_end) / span, np.percentile(wave_end, 10) / span))
for frac in (0.5, 0.7, 0.8, 0.9, 0.95):
    print("  at %.2f of the makespan %4d waves have finished for good" % (frac, int((wave_end < frac * span).sum())))
print("units that end last:")
for k in np.argsort(-t1)[:12]:
    print("  ends %.3f  starts %.3f  duration %.3f of the makespan, %2d passes, sub %2d, wave %d, block x %d y %d" % (t1[k] / span, t0[k] / span, d[k] / span, passes[k], sub[k], wave[k], bx[k], by[k]))
print("longest units:")
for k in np.argsort(-d)[:16]:
    print("  duration %.3f  starts %.3f  ends %.3f, %2d passes, sub %2d, block x %d y %d" % (d[k] / span, t0[k] / span, t1[k] / span, passes[k], sub[k], bx[k], by[k]))
