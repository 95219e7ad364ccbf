"""Frame times of smaller launches (the automatic engine takes the hybrid kernel below 9 blocks per wave):
  TUNE=K=V,.. python scripts/chunk_times.py   -- 1280x720 and 960x540 frames of the room, same camera and panning."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
torch.zeros(1, device="cuda")
info = sg.write_scene(os.environ.get("SCENE", "room"), "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); abi = M.hip_abi(); h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS); abi.set_stats(h, False)
for kv in [x for x in os.environ.get("TUNE", "").split(",") if x]:
    abi.set_tuning(h, kv.split("=")[0], float(kv.split("=")[1]))
for W, H in ((1280, 720), (960, 540)):
    buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    def frame(c):
        abi.render_chunk_device(h, binding.sensor(c, W, H), W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize()
        a, b = abi.kernel_times(h)
        return float(a[-1] + b[-1])
    same = [frame(sg.ROOM_CAMERA) for _ in range(20)]
    pan = []
    for f in range(1, 33):
        j = f % 16; tri = j if j <= 4 else (8 - j if j <= 12 else j - 16)
        c = list(sg.ROOM_CAMERA); c[4] += 2.0 * tri
        pan.append(frame(c))
    print("%dx%d: first %.2f ms, same frame %.3f ms, panning %.3f ms" % (W, H, same[0], sum(same[-8:]) / 8, sum(pan[-16:]) / 16))
