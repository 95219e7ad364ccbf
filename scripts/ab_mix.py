import os, sys, glob, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"]); flat = m.flatten()
info2 = sg.write_scene("room_nomtl", "/tmp/mt_scenes")
m2 = M.MythTracer(info2["obj"]); flat2 = m2.flatten()
for path in sorted(glob.glob(os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_*.so"))):
    name = os.path.basename(path)[len("libmythtracer_hip_"):-3]
    if name in ("diag", "prof"):
        continue
    abi = M.HipAbi(path)
    h = abi.scene_create(flat); abi.set_lights(h, sg.ROOM_LIGHTS)
    h2 = abi.scene_create(flat2); abi.set_lights(h2, sg.ROOM_LIGHTS)
    if os.environ.get("NO_STATS") and hasattr(abi.lib, "mt_scene_set_stats"):
        abi.set_stats(h, False); abi.set_stats(h2, False)
    out = []
    import hashlib, json
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["room_1920x1080_d5"]["sha256"]
    ok = "?"
    for hh, chunk, depth in ((h, None, 0), (h, (0, 0, 952, 1080), 5), (h, None, 5), (h2, None, 5)):
        rs = [abi.render_chunk(hh, sens, W, H, chunk=chunk, max_depth=depth) for _ in range(6)]
        out.append(min(r["stats"]["kernel_ms"] for r in rs[1:]))
        if hh is h and chunk is None and depth == 5:
            ok = "parity OK" if hashlib.sha256(rs[-1]["rgb"].tobytes()).hexdigest() == want else "PARITY MISMATCH"
    print("%-12s depth0 %.3f | left depth5 %.3f | full %.3f | primary-only %.3f | %s" % (name, *out, ok), flush=True)
