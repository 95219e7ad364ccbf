"""When do the work units of a launch start and end?  (-DMT_DEBUG_KNOBS build: MT_DEBUG_ITEM_CYCLES dumps per unit its
duration, and its start stamp and wave.)  Prints the launch's makespan in s_memtime ticks, the units that end last, the
waves' idle time at the end, and how much of the launch had fewer than all waves busy.

  WHAT=1080p|rank  RANK=5 WORLD=8  ENGINE=3  python scripts/unit_timeline.py"""
import ctypes, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "items_tl.bin")
os.environ["MT_DEBUG_ITEM_CYCLES"] = out
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding, tiling
torch.zeros(1, device="cuda")
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
abi = M.HipAbi(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_knobs.so")); h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS); abi.set_stats(h, False)
abi.set_engine(h, int(os.environ.get("ENGINE", "3")))
for k, v in [kv.split("=") for kv in os.environ.get("TUNE", "").split(",") if kv]:
    abi.set_tuning(h, k, float(v))
what = os.environ.get("WHAT", "rank")
if what == "1080p":
    W, H = 1920, 1080
    sens = binding.sensor(sg.ROOM_CAMERA, W, H)
    buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    n_items = (W // 8) * (H // 8)
    for i in range(10):
        abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
else:
    W, H, T = 3840, 2160, int(os.environ.get("TILE", "64"))
    world = int(os.environ.get("WORLD", "8")); rank = int(os.environ.get("RANK", "5"))
    sens = binding.sensor(sg.ROOM_CAMERA, W, H)
    f, s, n = tiling.rank_tiles(W, H, T, T, rank, world)
    n_items = n * (T // 8) * (T // 8)
    slots = torch.zeros(max(n, 1) * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda")
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
# s_memtime counts per XCD (eight unsynchronised counters; workgroups go to the XCDs round robin): every XCD's stamps
# are taken relative to its own first unit
xcd = (wave // 4) % 8
for x in range(8):
    sel = xcd == x
    if sel.any(): t0[sel] -= t0[sel].min()
t1 = t0 + d
span = t1.max()
print("%s engine %s: last launch %.3f + %.3f ms; %d units on %d waves; makespan %.3e ticks, sum of units %.3e = %.3f of makespan x 2048 waves" % (
    what, os.environ.get("ENGINE", "3"), a[-1], b[-1], len(d), len(np.unique(wave)), span, d.sum(), d.sum() / (span * 2048)))
ticks_per_ms = span / b[-1]
print("ticks per ms of the frame kernel: %.0f" % ticks_per_ms)
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
