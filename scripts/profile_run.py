"""Small workloads for the rocprofv3 passes of round 3 (python3 must follow `--` directly: no wrappers).

  python3 scripts/profile_run.py cold    8 first frames of the 1080p room frame (no measured costs: probe_kernel +
                                         mt::pool_kernel<false>), counters off, every frame's SHA checked
  python3 scripts/profile_run.py share   one rank's share (rank 5 of 8) of the 3840x2160 frame, 3 settling + 8 frames:
                                         mt::hybrid_kernel<false> (the automatic engine's choice for such a launch)
"""
import ctypes, hashlib, json, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mythtracer_amd as M
from mythtracer_amd import scenegen as sg, binding, tiling
what = sys.argv[1]
torch.cuda.init(); torch.zeros(1, device="cuda")
info = sg.write_scene("room", "/tmp/mt_scenes")
m = M.MythTracer(info["obj"])
abi = M.hip_abi(); h = abi.scene_create(m.flatten()); abi.set_lights(h, sg.ROOM_LIGHTS); abi.set_stats(h, False)
if what == "cold":
    W, H = 1920, 1080
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "frames.json")))["room_1920x1080_d5"]["sha256"]
    sens = binding.sensor(sg.ROOM_CAMERA, W, H)
    buf = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
    t = []
    for i in range(8):
        abi.set_scheduling(h, True)  # forget the costs: a first frame
        abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 5, ctypes.c_void_p(buf.data_ptr()))
        torch.cuda.synchronize()
        a, b = abi.kernel_times(h)
        t.append((float(a[-1]), float(b[-1])))
        assert hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest() == golden
    print(json.dumps({"workload": "room 1920x1080 d5, first frame of a geometry x 8", "order_ms": [x[0] for x in t], "frame_kernel_ms": [x[1] for x in t], "parity": "every frame identical to the reference's"}))
else:
    W, H, T = 3840, 2160, 64
    rank, world = 5, 8
    sens = binding.sensor(sg.ROOM_CAMERA, W, H)
    f, s, n = tiling.rank_tiles(W, H, T, T, rank, world)
    slots = torch.zeros(n * tiling.slot_bytes(T, T), dtype=torch.uint8, device="cuda")
    for i in range(11):
        abi.render_tiles_device(h, sens, W, H, T, T, f, s, n, 5, ctypes.c_void_p(slots.data_ptr()))
    torch.cuda.synchronize()
    a, b = abi.kernel_times(h)
    print(json.dumps({"workload": "room 3840x2160 d5, tiles k = 5 (mod 8), 11 frames", "order_ms": [float(x) for x in a], "frame_kernel_ms": [float(x) for x in b]}))
