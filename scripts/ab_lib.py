"""A/B of library variants in ONE process on the 1080p room frame: lib/libmythtracer_hip_<name>.so for every name in
argv (built by hand with another -D switch) against the current one; golden SHA check of every full frame."""
import hashlib, json, os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding
names = sys.argv[1:] + ["now"]
W, H = 1920, 1080
sens = binding.sensor(sg.ROOM_CAMERA, W, H)
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
flat = m.flatten()
want = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["room_1920x1080_d5"]["sha256"]
cases = [("full frame depth 5", None, 5), ("full frame depth 0", None, 0), ("left 952x1080 depth 5", (0, 0, 952, 1080), 5)]
for rep in range(2):
    for name in names:
        path = None if name == "now" else os.path.join(ROOT, "mythtracer_amd", "lib", "libmythtracer_hip_%s.so" % name)
        abi = M.HipAbi(path) if path else M.hip_abi()
        h = abi.scene_create(flat); abi.set_lights(h, sg.ROOM_LIGHTS)
        for engine in (1, 2):
            abi.set_engine(h, engine)
            for cname, chunk, depth in cases:
                rs = [abi.render_chunk(h, sens, W, H, chunk=chunk, max_depth=depth) for _ in range(6)]
                t = [r["stats"]["kernel_ms"] for r in rs]
                ok = ""
                if chunk is None and depth == 5:
                    ok = "parity OK" if hashlib.sha256(rs[-1]["rgb"].tobytes()).hexdigest() == want else "parity MISMATCH"
                print("%-4s engine %d %-24s cold %.2f warm min %.3f median %.3f ms %s" % (name, engine, cname, t[0], min(t[1:]), float(np.median(t[1:])), ok), flush=True)
        abi.lib.mt_scene_destroy(h)
