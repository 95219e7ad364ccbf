"""Warm / cold frame time of the 1080p room frame + parity against the golden SHA."""
import hashlib, json, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); m.set_lights(sg.ROOM_LIGHTS)
abi = M.hip_abi(); h = m.device_scene(); abi.set_lights(h, sg.ROOM_LIGHTS)
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
want = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["room_1920x1080_d5"]["sha256"]
modes = [int(x) for x in os.environ.get("MODES", "0").split(",")]
for mode in modes:
    abi.set_traversal_mode(h, mode)
    abi.set_scheduling(h, True)
    t = []
    for i in range(int(os.environ.get("FRAMES", "8"))):
        r = abi.render_chunk(h, sens, W, H)
        t.append(r["stats"]["kernel_ms"])
    sha = hashlib.sha256(r["rgb"].tobytes()).hexdigest()
    print("mode %d: cold %.2f ms, warm min %.2f median %.2f ms, parity %s" % (
        mode, t[0], min(t[1:]), float(np.median(t[1:])), "OK" if sha == want else "MISMATCH"), flush=True)
