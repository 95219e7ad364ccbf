"""Time line of ONE wave's hit-set walk (diagnostic -DMT_PROF build, MT_DEBUG_TIMELINE): cycles between the stamps
of mt_trace.h (MT_TL), per kind of transition."""
import os, sys, collections, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "gpurun_out", "timeline.bin")
os.environ["MT_DEBUG_TIMELINE"] = out
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
abi = M.HipAbi(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_prof.so"))
h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS)
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
abi.set_engine(h, 1)
for _ in range(3):
    r = abi.render_chunk(h, sens, W, H)
a = np.fromfile(out, dtype=np.uint64)
n = int(min(a[0], len(a) - 1))
st = a[1:1 + n]
st = st[st != 0]; n = len(st)
t = (st >> np.uint64(8)).astype(np.int64); tag = (st & np.uint64(0xff)).astype(np.int64)
order = np.argsort(t, kind="stable"); t, tag = t[order], tag[order]
names = {1: "walk begins", 2: "ENTER", 3: "record landed", 4: "child tests done", 5: "copies issued", 6: "own list done",
         7: "frame open + short leaves", 8: "back at a frame", 9: "offer done / next child"}
print("kernel_ms", r["stats"]["kernel_ms"], "stamps", n)
d = collections.defaultdict(list)
for i in range(1, n):
    if tag[i] == 1: continue
    d[(tag[i - 1], tag[i])].append(t[i] - t[i - 1])
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    print("%-28s -> %-28s n %5d  mean %7.0f  median %7.0f  p90 %7.0f  share %.3f" % (names.get(k[0], k[0]), names.get(k[1], k[1]), len(v), v.mean(), np.median(v), np.percentile(v, 90), v.sum() / tot))
