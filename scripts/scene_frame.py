"""Warm / panning / first-frame times of one scene's 1920x1080 frame, optionally through another build of the library.

  SCENE=loft LIB=hs15 python scripts/scene_frame.py     (lib/libmythtracer_hip_hs15.so: -DMT_HS_MAX_DEPTH=15, the ordered
                                                         descent on a 16-level tree)"""
import ctypes, hashlib, json, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
torch.zeros(1, device="cuda")
scene = os.environ.get("SCENE", "loft")
info = sg.write_scene(scene, "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
lib = os.environ.get("LIB")
abi = M.HipAbi(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_%s.so" % lib)) if lib else M.hip_abi()
h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS); abi.set_stats(h, False)
for kv in [x for x in os.environ.get("TUNE", "").split(",") if x]:
    abi.set_tuning(h, kv.split("=")[0], float(kv.split("=")[1]))
W, H = 1920, 1080
frames = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))
gold = frames.get("%s_%dx%d_d5" % (scene, W, H), {}).get("sha256")
buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
order_ms = []
def frame(c):
    abi.render_chunk_device(h, binding.sensor(c, W, H), W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
    torch.cuda.synchronize()
    a, b = abi.kernel_times(h)
    order_ms.append(float(a[-1]))
    return float(a[-1] + b[-1])
cold = frame(sg.ROOM_CAMERA)
ok = hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest() == gold if gold else None
warm = [frame(sg.ROOM_CAMERA) for _ in range(24)]
pan = []
for f in range(1, 33):
    j = f % 16; tri = j if j <= 4 else (8 - j if j <= 12 else j - 16)
    c = list(sg.ROOM_CAMERA); c[4] += 2.0 * tri
    pan.append(frame(c))
print("   work-order kernels of the panning frames: %.3f ms" % (sum(order_ms[-16:]) / 16))
if os.environ.get("PERFRAME"): print("   panning frames (ms): " + " ".join("%.2f" % x for x in pan[-16:]))
frame(sg.ROOM_CAMERA)  # (scheduled from the last panning frame's costs)
ok2 = hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest() == gold if gold else None
if ok2 is False: print("MISMATCH of the frame after the pan")
abi.set_stats(h, True); abi.read_stats(h); frame(sg.ROOM_CAMERA); st = abi.read_stats(h)
rays = st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
print("%s%s: %d triangles; first frame %.2f ms (%s), warm %.3f ms, panning %.3f ms = %.0f Mray/s (%d rays per frame, %.1f node visits per ray, %d wave steps)" % (
    scene, " [%s]" % lib if lib else "", info["triangles"], cold, "golden ok" if ok else ("MISMATCH" if ok is False else "no golden"),
    sum(warm[-16:]) / 16, sum(pan[-16:]) / 16, rays / (sum(pan[-16:]) / 16 * 1e-3) / 1e6, rays, st["node_visits"] / rays, st["wave_node_steps"]))
