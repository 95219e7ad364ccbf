#!/usr/bin/env python3
"""bench.py — the hot path on BASELINE.json's headline configurations.

One "step" = one frame through MythTracer::RayTrace's pixel loop on the GPU
(mt_render_chunk_device / mt_render_tiles_device of the C ABI).  Scene, lights
and sensor are resident in HBM before the timed region; the frame stays in HBM
(the PCIe-inclusive rate is in DESIGN.md).

  python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment bench.py starts its own
ranks: the parent -- before it imports torch or touches HIP -- runs
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1 ... bench.py <same arguments>` as a child process and relays its
output and exit code.  Under a launcher (WORLD_SIZE set) it is one rank.

Workloads (the ~100k-triangle room of mythtracer_amd/scenegen.py stands in for
the unavailable living-room model; 3 lights with shadow rays, the reference's
MAX_RECURSION_LEVEL = 5):

  N = 1   BASELINE configs[2]: 1920x1080, one launch for the whole frame.
  N > 1   BASELINE configs[4]: 3840x2160, STRONG scaling: every rank holds a
          scene replica (as every worker does in the reference,
          main_net_worker.cc:29-32), renders ITS 64x64 tiles of the ONE frame, and
          the tile buffers are gathered to rank 0 over RCCL and blitted into the
          frame (main_net_master.cc:223-236) -- the reference's only exchange
          step.  Which tiles: dealt out by cost (--ownership dealt, the default:
          every rank orders the tiles by the all-reduced cost map of the previous
          frame and takes its deal -- the static counterpart of the master's pull
          queue, main_net_master.cc:62-80; include/mythtracer_hip.h,
          mt_order_tiles_device) or k = rank (mod N) (--ownership modular).
          `--scaling weak` grows the frame with N instead (16k x 9k pixels,
          k = round(120 sqrt(N))).
  --width/--height/--max-depth/--scene select the other BASELINE configurations.

Regime of the timed steps (--regime):
  moving  (default) the reference's own loop, main_local.cc:51-76, turns the
          camera 2 degrees per frame: every frame is a NEW frame, scheduled from
          the previous frame's block costs re-projected through the camera
          change.  Here the camera pans in 2-degree steps within +-8 degrees of
          the golden camera (yaw0 + 0, 2, .. 8, 6, .. -8, -6, .. 0, ...), so that
          the workload stays the frame BASELINE.json names whatever K is (a
          monotonic 256-degree turn averages over easier views: 3.6 ms per
          frame), and ends ON the golden camera: the last timed frame's SHA-256
          is compared with the frame the compiled reference rendered.  Ray
          counts come from an untimed pass over the same K frames with the work
          counters on, in which a dozen frames are also crop-checked against the
          oracle.
  warm    every step re-renders the golden camera's frame (round 1/2's headline;
          reported as extras.warm_same_frame in the default run).

Rank 0 prints ONE JSON line; the process exits non-zero if a frame does not
match the reference's.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
N_SIMD = 1024          # 256 CUs x 4 SIMDs
RAY_KEYS = ("rays_primary", "rays_secondary", "rays_shadow")
KERNEL_SOURCES = ("mythtracer_amd/csrc/mt_capi.hip", "mythtracer_amd/csrc/mt_render.hip", "mythtracer_amd/csrc/mt_pool.h",
                  "mythtracer_amd/csrc/mt_trace.h", "mythtracer_amd/csrc/mt_shade.h", "mythtracer_amd/csrc/mt_device.h",
                  "mythtracer_amd/csrc/mt_queues.h", "mythtracer_amd/csrc/mt_order.h", "include/mythtracer_hip.h")


def kernel_source_sha256() -> str:
    """Identity of the kernels a profile was measured on: SHA-256 over the HIP sources."""
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def reference_bytes(c: dict, pixels: int) -> int:
    """bytes the REFERENCE's un-pruned algorithm touches per frame —
    48 B per node/child box test, 48 B per triangle pre-filter test, 72 B per
    Möller–Trumbore test, vertices+normals+material per shaded hit, 3 B/pixel."""
    return (48 * c["box_tests"] + 48 * c["tri_tests"] + 72 * c["mt_tests"] +
            (72 + 72 + 136) * c["shaded_hits"] + 3 * pixels)


def cpu_baseline(scene_obj, cam, lights, W, H, chunk, max_depth, rays_in_chunk):
    """Times the reference itself (oracle/_ref, compiled from /root/reference in
    the build container) on the host cores; falls back to our CPU port when the
    prebuilt reference is absent or the depth is not the reference's compile-time 5."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    cores = os.cpu_count() or 1
    whole = chunk[2] == W and chunk[3] == H
    sample = ("the whole %dx%d frame" % (W, H) if whole else
              "chunk x=%d y=%d %dx%d of the %dx%d frame" % (*chunk, W, H)) + ", all host threads (OpenMP rows)"
    if orclib.have_ref() and max_depth == 5:
        with tempfile.TemporaryDirectory() as td:
            r = orclib.run_ref(td, scene_obj, (W, H), chunk=chunk, cam=cam, lights=lights)
        if r.get("returncode") == 0:
            sec = r["time"]["seconds"]
            return {"value": rays_in_chunk / sec / 1e6, "unit": "Mray/s",
                    "cores": int(r["time"]["threads"]), "kind": "reference",
                    "sample": sample, "seconds": sec}
    o = orclib.OracleScene(scene_obj)
    o.set_lights(lights)
    r = o.render(cam, W, H, chunk=chunk, max_level=max_depth)
    rays = sum(r["counters"][k] for k in RAY_KEYS)
    return {"value": rays / r["seconds"] / 1e6, "unit": "Mray/s", "cores": cores,
            "kind": "port", "sample": sample, "seconds": r["seconds"]}


def self_launch(n_gpus: int) -> int:
    """--gpus N without a launcher: start the ranks as a CHILD process (this
    parent has not touched torch or HIP) and relay its output and exit code."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("bench.py: --gpus %d without a launcher; starting %s" % (n_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--width", type=int, default=0, help="default: 1920 at N = 1, 3840 at N > 1")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--regime", choices=("moving", "warm"), default="moving")
    ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--scene", default="room")
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--ownership", choices=("dealt", "modular"), default="dealt",
                    help="N > 1: tiles dealt out by the previous frame's costs (default) or k = rank (mod N)")
    ap.add_argument("--engine", type=int, default=0, help="0 automatic (default), 1 state machine, 2 ray pool, 3 hybrid")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", choices=("full", "band"), default="full",
                    help="reference timed on the whole frame (about 30 s) or on a quarter-frame band")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras (cold frame, same-frame regime, reference-work counts, crop checks)")
    ap.add_argument("--crop-checks", type=int, default=12, help="frames of the counting pass checked on a crop against the oracle")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))  # before anything imports torch / touches the GPU

    import numpy as np
    import torch
    import torch.distributed as dist
    import mythtracer_amd as M
    from mythtracer_amd import multi, scenegen, tiling
    from mythtracer_amd.binding import sensor as host_sensor

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d; the launcher's world size is used" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    # Rehearsal on a one-GPU box (MT_BENCH_EMULATE_RANKS=1): every rank uses GPU 0
    # and the exchange goes through gloo on host copies; everything else is the
    # code the real N-GPU run executes.  Never used by the driver.
    emulate = os.environ.get("MT_BENCH_EMULATE_RANKS") == "1" and world > 1
    if emulate:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if world > 1:  # before anything else touches the device
        if emulate:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        backend = dist.get_backend()
        world = dist.get_world_size()
    xdev = torch.device("cpu") if emulate else dev  # where collective operands live

    if args.width > 0 and args.height > 0:
        W, H = args.width, args.height
        scaling = "strong"
    elif world == 1:
        W, H = 1920, 1080
        scaling = "strong"
    elif args.scaling == "weak":
        k = int(round(120.0 * (world ** 0.5)))
        W, H = 16 * k, 9 * k          # N=4: 3840x2160
        scaling = "weak"
    else:
        W, H = 3840, 2160             # BASELINE configs[4]
        scaling = "strong"
    scene_dir = os.path.join(tempfile.gettempdir(), "mt_bench_scene_%d_%d" % (os.getuid(), rank))
    info = scenegen.write_scene(args.scene, scene_dir)
    cam, lights = scenegen.ROOM_CAMERA, scenegen.ROOM_LIGHTS

    t_load = time.time()
    mt = M.MythTracer(info["obj"], device=local_rank)
    mt.set_lights(lights)
    h = mt.device_scene()          # finalize + upload (HBM-resident from here on)
    abi = M.hip_abi()
    abi.set_lights(h, lights)
    abi.set_engine(h, args.engine)
    t_load = time.time() - t_load
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ---- the cameras of the run: K timed steps preceded by W warm-up steps, one continuous turn of 2 degrees
    # per frame that ENDS on the golden camera (moving), or the golden camera every time (warm)
    K, Wu = args.steps, args.warmup

    def cam_of(offset_steps):
        """offset_steps <= 0 = frames before the last timed one; a triangular wave of amplitude 4 steps (8 degrees)"""
        c = list(cam)
        if args.regime == "moving":
            j = (-offset_steps) % 16
            tri = j if j <= 4 else (8 - j if j <= 12 else j - 16)
            c[4] = cam[4] + 2.0 * tri  # yaw, degrees (Camera{origin, pitch, yaw, roll, aov})
        return c
    cams_timed = [cam_of(i - (K - 1)) for i in range(K)]
    cams_warm = [cam_of(j - (K - 1) - Wu) for j in range(Wu)]
    sens_timed = [host_sensor(c, W, H) for c in cams_timed]
    sens_warm = [host_sensor(c, W, H) for c in cams_warm]
    sens = host_sensor(cam, W, H)  # the golden camera (== sens_timed[-1])

    tw = th = args.tile
    first, stride, n_mine = tiling.rank_tiles(W, H, tw, th, rank, world)
    n_max = tiling.max_tiles_per_rank(W, H, tw, th, world)   # equal slot counts on every rank
    frame = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    dealt = world > 1 and args.ownership == "dealt" and tw % 8 == 0 and th % 8 == 0
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    if dealt:
        # cost-balanced ownership: the order of the tiles (a function of the all-reduced cost map: the same on every
        # rank), this rank's list, and on rank 0 every rank's list for the blit
        n_tiles_total = tiling.tile_grid(W, H, tw, th)[0] * tiling.tile_grid(W, H, tw, th)[1]
        n_of = [abi.dealt_tile_count(W, H, tw, th, world, r) for r in range(world)]
        n_mine = n_of[rank]
        order = torch.zeros(n_tiles_total, dtype=torch.int32, device=dev)
        lists = [torch.zeros(max(n_of[r], 1), dtype=torch.int32, device=dev) for r in (range(world) if rank == 0 else [rank])]
        my_list = lists[rank] if rank == 0 else lists[0]
        deal_state = {"have_map": False, "list_id": 0, "last_sensor": None, "at_rest": 0}
    if world > 1:
        mine = torch.zeros(n_max * tiling.slot_bytes(tw, th), dtype=torch.uint8, device=dev)
        gathered = ([torch.zeros_like(mine, device=xdev) for _ in range(world)] if rank == 0 else None)
        # the ranks' block costs of the frame just rendered, 4 bytes per 8x8 block of the whole frame: a moving
        # camera's next frame is ordered by them (include/mythtracer_hip.h, mt_scene_export_costs_device)
        map_w, map_h = (W + 7) // 8, (H + 7) // 8
        cost_map = torch.zeros((map_h, map_w), dtype=torch.int32, device=dev)
        exchange_costs = (args.regime == "moving" or dealt) and tw % 8 == 0 and th % 8 == 0

    def E():
        return torch.cuda.Event(enable_timing=True)
    ev = [(E(), E(), E()) for _ in range(K)]  # before render, after render, after gather + blit

    def step(s12, i=None, out=None):
        out = frame if out is None else out
        if i is not None:
            ev[i][0].record()
        if world == 1:
            abi.render_chunk_device(h, s12, W, H, (0, 0, W, H), args.max_depth,
                                    ctypes.c_void_p(out.data_ptr()), None, stream)
        elif dealt:
            # new lists whenever the camera moved (and for the first two frames of a camera at rest: by number, then by
            # the first measured costs); a camera at rest keeps its lists -- and with them the per-slot cost history
            st = deal_state
            key = bytes(np.asarray(s12, dtype=np.float64).tobytes())
            st["at_rest"] = st["at_rest"] + 1 if st["last_sensor"] == key else 0
            st["last_sensor"] = key
            if st["list_id"] == 0 or st["at_rest"] < 2:
                st["list_id"] += 1
                if st["have_map"]:
                    abi.order_tiles_device(h, vp(cost_map), map_w, map_h, W, H, tw, th, vp(order), stream)
                for k, r in enumerate(range(world) if rank == 0 else [rank]):
                    abi.deal_tiles_device(h, vp(order) if st["have_map"] else None, W, H, tw, th, world, r, vp(lists[k]), stream)
            abi.render_tile_list_device(h, s12, W, H, tw, th, vp(my_list), n_mine, st["list_id"], args.max_depth, vp(mine), stream)
        else:
            abi.render_tiles_device(h, s12, W, H, tw, th, first, stride, n_mine, args.max_depth,
                                    ctypes.c_void_p(mine.data_ptr()), stream)
        if i is not None:
            ev[i][1].record()
        if world > 1:
            def blit(slots, f_r, s_r, n_r):
                slots = slots.to(dev)  # no-op except in the rehearsal mode
                abi.blit_tiles_device(h, W, H, tw, th, f_r, s_r, n_r, ctypes.c_void_p(slots.data_ptr()),
                                      ctypes.c_void_p(out.data_ptr()), stream)
                if emulate:
                    torch.cuda.synchronize()  # `slots` is a temporary here

            def blit_list(slots, r):
                slots = slots.to(dev)
                abi.blit_tile_list_device(h, W, H, tw, th, vp(lists[r]), n_of[r], vp(slots), vp(out), stream)
                if emulate:
                    torch.cuda.synchronize()
            if dealt:
                multi.gather_and_blit_lists(dist, mine.to(xdev), gathered, rank, world, lambda r: r, blit_list)
            else:
                multi.gather_and_blit(dist, mine.to(xdev), gathered, rank, world, W, H, tw, th, blit)
            if exchange_costs:
                # second exchange step: every rank's costs into one map (element-wise MAX: a block belongs to one rank)
                cost_map.zero_()
                abi.export_costs_device(h, ctypes.c_void_p(cost_map.data_ptr()), map_w, map_h, stream)
                if emulate:
                    cm = cost_map.cpu()
                    dist.all_reduce(cm, op=dist.ReduceOp.MAX)
                    cost_map.copy_(cm)
                else:
                    dist.all_reduce(cost_map, op=dist.ReduceOp.MAX)
                abi.import_costs_device(h, ctypes.c_void_p(cost_map.data_ptr()), map_w, map_h, stream)
                if dealt:
                    deal_state["have_map"] = True
        if i is not None:
            ev[i][2].record()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def golden_for(Wg, Hg):
        """(sha256 of the reference's frame for the golden camera, who made it, how to compare) or (None, ..)"""
        gpath = os.path.join(ROOT, "tests", "golden", "frames.json")
        if not os.path.exists(gpath):
            return None, None, None
        frames = json.load(open(gpath))
        key = "%s_%dx%d_d%d" % (args.scene, Wg, Hg, args.max_depth)
        if key in frames:
            return frames[key]["sha256"], frames[key].get("made_by", "the compiled reference"), "full"
        if (Wg, Hg) == (3840, 2160):
            # Sensor::GetRay divides the same corner vectors by W and H: the ray of 4K pixel (2x, 2y) is
            # bit-for-bit that of 1080p pixel (x, y), so the even pixels of this frame must be the
            # reference's 1080p frame
            k2 = "%s_1920x1080_d%d" % (args.scene, args.max_depth)
            if k2 in frames:
                return frames[k2]["sha256"], frames[k2].get("made_by", "the compiled reference"), "even"
        return None, None, None
    golden, golden_from, golden_mode = golden_for(W, H)

    def verdict(buf):
        """(sha of the buffer, parity text or None, mismatch flag) for a frame of the GOLDEN camera."""
        img = buf.cpu().numpy()
        sha = hashlib.sha256(img.tobytes()).hexdigest()
        if golden is None:
            return sha, None, False
        if golden_mode == "even":
            cmp_sha = hashlib.sha256(np.ascontiguousarray(img[::2, ::2]).tobytes()).hexdigest()
            text = "even pixels identical to the 1920x1080 golden frame"
        else:
            cmp_sha, text = sha, "frame identical to the reference's"
        if golden_from != "the compiled reference":
            text += " (golden made by: %s)" % golden_from
        bad = cmp_sha != golden
        return sha, ("MISMATCH vs golden frame" if bad else text), bad

    mismatch = False
    # ---- 1. warm-up and the COUNTING pass (untimed, work counters on): the rays of every timed frame, and a
    # dozen of the frames checked on a crop against the oracle
    for s12 in sens_warm:
        step(s12)
    fence()
    abi.read_stats(h)              # drop warm-up counts
    keys = ["rays_primary", "rays_secondary", "rays_shadow", "box_tests", "node_visits",
            "tri_tests", "mt_tests", "shaded_hits", "wave_node_steps", "wave_tri_steps",
            "bytes_scalar", "bytes_vector"]
    tot = {k: 0 for k in keys}
    last_counters = None
    orc = None
    crops_ok, crops_n = True, 0
    n_checks = 0 if (args.no_extras or rank != 0 or args.regime != "moving") else min(args.crop_checks, K)
    check_at = set(int(round(j * (K - 1) / max(n_checks - 1, 1))) for j in range(n_checks))
    if n_checks:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orclib
        orc = orclib.OracleScene(info["obj"])
        orc.set_lights(lights)
    elapsed_counting = 0.0
    for i in range(K):
        t1 = time.perf_counter()
        step(sens_timed[i])
        torch.cuda.synchronize()
        elapsed_counting += time.perf_counter() - t1
        last_counters = abi.read_stats(h)  # this rank, this frame
        for k in keys:
            tot[k] += last_counters[k]
        if i in check_at:
            cw, ch = 96, 48
            cx, cy = (211 * i) % (W - cw), (H // 3 + 37 * i) % (H - ch)
            want = orc.render(cams_timed[i], W, H, chunk=(cx, cy, cw, ch), max_level=args.max_depth)["rgb"]
            got = frame[cy:cy + ch, cx:cx + cw].cpu().numpy()
            crops_ok = crops_ok and bool(np.array_equal(got, want))
            crops_n += 1
    elapsed_counting /= max(K, 1)
    mismatch = mismatch or not crops_ok
    abi.kernel_times(h)
    # ---- 2. the TIMED steps: the kernels built without the work counters (mt_scene_set_stats: counting costs
    # about a tenth of a frame).  The camera jumps back to the start of the turn: the warm-up frames come again.
    abi.set_stats(h, False)
    for s12 in sens_warm:
        step(s12)
    if Wu == 0:
        step(sens_timed[0])          # first launch of the counter-free kernels
    fence()
    abi.kernel_times(h)            # drop the durations so far
    t0 = time.perf_counter()
    for i in range(K):
        step(sens_timed[i], i)
    fence()
    elapsed = time.perf_counter() - t0
    # the frame the LAST timed step wrote (the golden camera, counter-free kernels): hashed before anything
    # else renders into `frame`
    sha_timed, parity_timed, bad = (None, None, False)
    if rank == 0:
        sha_timed, parity_timed, bad = verdict(frame)
    mismatch = mismatch or bad
    # per-kernel device durations of the timed steps: HIP events the library
    # records on the launch stream around each of its kernels (at most the last 64)
    k_order, k_frame = abi.kernel_times(h, 64)

    vec = torch.tensor([float(tot[k]) for k in keys], dtype=torch.float64, device=xdev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
    render_ms = [a.elapsed_time(b) for a, b, _ in ev]
    exchange_ms = [b.elapsed_time(c) for _, b, c in ev]
    kmax = torch.tensor([float(k_frame.mean()) if len(k_frame) else 0.0,
                         float(k_order.mean()) if len(k_order) else 0.0,
                         sum(render_ms) / max(len(render_ms), 1),
                         sum(exchange_ms) / max(len(exchange_ms), 1)], dtype=torch.float64, device=xdev)
    if world > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    tot = {k: int(v) for k, v in zip(keys, vec.tolist())}   # all ranks, all K timed frames
    rays = tot["rays_primary"] + tot["rays_secondary"] + tot["rays_shadow"]
    per_frame = {k: tot[k] / max(K, 1) for k in keys}        # mean over the K frames

    out = None
    if rank == 0:
        k_ms, k_order_ms, k_step_ms, k_exchange_ms = (float(x) for x in kmax.tolist())
        roof = None
        if world == 1:
            # ---- roofline of the dominant kernel (the frame kernel).  What binds it is VALU instruction issue (no
            # contraction -> no MFMA; 33 MB working set -> caches, not HBM): `achieved` = SIMD cycles per second in
            # which a VALU instruction issues = VALU-busy cycles per launch (rocprofv3 PMC pass, stamped with the
            # kernel sources it was measured on) / this run's live kernel duration; `peak` = 1024 SIMDs x clock.
            # Stale numbers are not reported: when the kernel sources differ from the profile's, the PMC-derived
            # fields are null.  The byte rates (requested bytes from the kernel's own counters -- live --, memory-side
            # traffic from the PMC pass) are labelled extras.
            req = (tot["bytes_scalar"] + tot["bytes_vector"]) / max(K, 1)
            src_sha = kernel_source_sha256()
            pmc, pmc_file = None, os.path.join("profiles", "pmc_frame_kernel.json")
            if os.path.exists(os.path.join(ROOT, pmc_file)):
                allp = json.load(open(os.path.join(ROOT, pmc_file)))
                e = allp.get("%s_%dx%d_d%d_%s" % (args.scene, W, H, args.max_depth, args.regime))
                if isinstance(e, dict) and e.get("kernel_source_sha256") == src_sha:
                    pmc = e
            roof = {"bound": "valu_issue", "achieved": None, "peak": None, "unit": "G SIMD-cycles/s with a VALU instruction issuing",
                    "frac": None, "traffic": None,
                    "kernel": "mt::render_kernel<false> (throughput engine) / mt::pool_kernel<false> (latency engine)",
                    "kernel_ms": k_ms, "launches_averaged": int(len(k_frame)),
                    "other_kernels_ms": {"work order (forecast / schedule / probe / primary)": k_order_ms},
                    "step_ms_device": k_step_ms,
                    "kernel_source_sha256": src_sha}
            if pmc:
                busy = pmc["valu_busy_simd_cycles_per_launch"]       # SQ_ACTIVE_INST_VALU x 4
                clock = pmc["effective_clock_GHz"]
                ach = busy / (k_ms * 1e-3) / 1e9
                roof.update({"achieved": ach, "peak": N_SIMD * clock, "frac": ach / (N_SIMD * clock),
                             "traffic": pmc.get("hbm_bytes_per_launch"),
                             "lane_weighted_frac": ach / (N_SIMD * clock) * pmc["active_lanes_per_valu_instruction"] / 64.0,
                             "pmc": {k2: pmc[k2] for k2 in pmc},
                             "pmc_source": {"file": pmc_file, "measured_at_commit": pmc.get("commit"),
                                            "note": "rocprofv3 --pmc passes of this bench command (scripts/pmc_passes.sh); used "
                                                    "because the kernel sources are the ones it was measured on"}})
            else:
                roof["pmc_source"] = {"file": pmc_file, "note": "no PMC profile of THESE kernel sources and this workload: "
                                                               "the counter-derived fields are null rather than stale"}
            roof["hbm"] = {"peak_GBps": HBM_PEAK_GBS,
                           "requested_GBps": req / (k_ms * 1e-3) / 1e9, "requested_frac": req / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "requested_bytes_per_launch": {"scalar_wave_uniform": per_frame["bytes_scalar"],
                                                          "vector_per_lane": per_frame["bytes_vector"]},
                           "memory_side_GBps": (pmc["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9) if pmc and pmc.get("hbm_bytes_per_launch") else None,
                           "note": "requested = bytes the frame kernel asks for per launch (its own counters, live; mostly "
                                   "served by scalar cache, L1, L2); memory_side = FETCH_SIZE x2 + WRITE_SIZE of the PMC pass"}

        out = {
            "metric": "Mray/s (primary+shadow+secondary; ray = one OctTree::IntersectRay)",
            "value": rays / elapsed / 1e6, "unit": "Mray/s", "n_gpus": world,
            "steps": K, "warmup": Wu,
            "ms_per_step": elapsed / max(K, 1) * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "value_regime": args.regime,
            "config": {"workload": "%s scene (%d triangles, synthetic stand-in for the living-room "
                                   ".obj), %dx%d, %d lights with shadow rays, max recursion %d%s"
                                   % (args.scene, info["triangles"], W, H, len(lights), args.max_depth,
                                      " = BASELINE configs[2]" if (W, H, args.max_depth, world, args.scene) == (1920, 1080, 5, 1, "room")
                                      else (" = BASELINE configs[4]" if (W, H, args.max_depth, args.scene) == (3840, 2160, 5, "room") and world > 1
                                            else "")),
                       "tile": ("%dx%d tiles, %s, over %d ranks, gathered to rank 0 (%s) and blitted" % (
                                    tw, th, "dealt out by the all-reduced cost map of the previous frame (most expensive first, rounds of "
                                    "alternating direction)" if dealt else "tile k to rank k mod N", world, backend))
                               if world > 1 else "whole frame per launch, 8x8-pixel work items",
                       "regime": ("moving camera: yaw changes by 2 degrees per step as in the reference's loop (main_local.cc:51-76), "
                                  "panning within +-8 degrees of the golden camera; every step is a NEW frame scheduled from the "
                                  "previous frame's block costs re-projected through the camera change; the last step is the "
                                  "golden camera (extras.warm_same_frame / cold_frame_ms give the other regimes)") if args.regime == "moving" else
                                 "warm: every timed step re-renders the golden camera's frame, scheduled from its own measured block costs",
                       "engine": {0: "automatic", 1: "state machine", 2: "ray pool", 3: "hybrid"}[args.engine],
                       "scene_sha256": info["sha256"]},
            "frame_ms_wall": elapsed / max(K, 1) * 1e3,
            "frame_ms_wall_with_work_counters_and_a_sync_per_frame": elapsed_counting * 1e3,
            "work_counters": "off in the timed steps (mt_scene_set_stats(scene, 0)); ray counts and requested bytes "
                             "are those of an untimed pass over the SAME K frames rendered with the counters on",
            "render_ms_device": k_step_ms,
            "rays_per_frame_mean": {k: per_frame[k] for k in RAY_KEYS},
            "rays_last_frame": {k: last_counters[k] for k in RAY_KEYS} if world == 1 else None,
            "Mray_s_primary_plus_shadow": (tot["rays_primary"] + tot["rays_shadow"]) / elapsed / 1e6,
            "frame_sha256": sha_timed,
            "parity": parity_timed,
            "parity_scope": "the frame the LAST TIMED step wrote (counter-free kernels, golden camera), hashed right after the timed loop"
                            + ("; %d frames of the turn crop-checked against the oracle: %s" % (crops_n, "all equal" if crops_ok else "MISMATCH") if crops_n else ""),
            "scene_load_s": t_load,
            "roofline": roof,
        }
        if world > 1:
            out["ranks_reported_by_backend"] = world
            out["exchange_ms_device"] = {"gather_plus_blit_rank0_plus_cost_map_all_reduce": k_exchange_ms,
                                         "bytes_gathered": int(n_max * tiling.slot_bytes(tw, th) * world),
                                         "cost_map_bytes_all_reduced": int(map_w * map_h * 4) if exchange_costs else 0}

    # ---- untimed extras, one GPU only
    if world == 1 and rank == 0 and not args.no_extras:
        extras = {}
        scratch = torch.zeros_like(frame)
        sp = ctypes.c_void_p(scratch.data_ptr())

        def timed_frame(sensor12):
            abi.render_chunk_device(h, sensor12, W, H, (0, 0, W, H), args.max_depth, sp, None, stream)
            torch.cuda.synchronize()
            a, b = abi.kernel_times(h)
            return float(a[-1] + b[-1])

        # (0) what the DROP-IN CALLER waits for: the frame loop of main_local.cc:51-132 through the facade -- lights pushed
        # again, a Camera, RayTrace(W, H, &cam, &bitmap) into one vector, per frame -- wall-clocked around each RayTrace
        # call: light upload (skipped when unchanged), work order, frame kernel, the copy of the frame into the caller's
        # vector (through page-locked staging, in pieces under each other's DMA), the device status.  Same panning regime, the
        # last frame is the golden camera's and is compared with the reference's.
        try:
            mt.frame_loop(cam, W, H, 20, 2.0 if args.regime == "moving" else 0.0)   # warm-up: registration, first frames
            e2e_ms, e2e_rgb = mt.frame_loop(cam, W, H, 32, 2.0 if args.regime == "moving" else 0.0)
            ka, kb = abi.kernel_times(h, 32)
            _, p_e2e, bad = verdict(torch.from_numpy(e2e_rgb))
            mismatch = mismatch or bad
            out["frame_ms_end_to_end"] = float(e2e_ms.mean())
            extras["end_to_end"] = {"frames": int(len(e2e_ms)), "ms_mean": float(e2e_ms.mean()), "ms_min": float(e2e_ms.min()),
                                    "ms_max": float(e2e_ms.max()), "kernels_ms_mean": float((ka + kb).mean()),
                                    "beyond_the_kernels_ms": float(e2e_ms.mean() - (ka + kb).mean()), "parity": p_e2e,
                                    "what": "MythTracer::RayTrace(W, H, &cam, &bitmap) per frame as main_local.cc calls it "
                                            "(host_capi.cc mth_frame_loop): lights, sensor, launch, D2H into the caller's vector, sync"}
        except Exception as e:  # the bench line must still come out
            extras["end_to_end"] = {"error": repr(e)}
        # (1) cold frame: no cost history (first frame of a geometry); counter-free kernels like the timed
        # steps; the frame it wrote is compared with the golden too
        abi.set_stats(h, False)
        abi.set_scheduling(h, True)  # forgets the recorded costs
        extras["cold_frame_ms"] = timed_frame(sens)
        _, p_cold, bad = verdict(scratch)
        extras["cold_frame_parity"] = p_cold
        mismatch = mismatch or bad
        # (2) the same frame again and again (rounds 1-2's headline regime)
        same = [timed_frame(sens) for _ in range(24)]
        _, p_same, bad = verdict(scratch)
        mismatch = mismatch or bad
        extras["warm_same_frame"] = {"frames": len(same), "ms_mean_last_16": sum(same[-16:]) / 16.0, "ms_min": min(same),
                                     "ms_first_after_cold": same[0], "parity": p_same,
                                     "Mray_s": sum(last_counters[k] for k in RAY_KEYS) / (sum(same[-16:]) / 16.0 * 1e-3) / 1e6}
        extras["work_counters"] = "off for cold_frame_ms and warm_same_frame, as in the timed steps"
        # (rounds 1-2 reported this regime as `value`: kept at top level so that the rounds stay comparable)
        out["value_warm"] = extras["warm_same_frame"]["Mray_s"]
        # (3) the same frame with the counters ON must be the same bytes (the kernels differ only in the counting)
        abi.set_stats(h, True)
        abi.read_stats(h)
        timed_frame(sens)
        sha_on, p_on, bad = verdict(scratch)
        mismatch = mismatch or bad or (args.regime == "warm" and sha_on != sha_timed)
        extras["counters_on_frame_parity"] = p_on
        # (4) the work of the REFERENCE's un-pruned traversal on this frame (traversal
        # mode 7 visits every subtree the reference visits; the image is the same)
        abi.read_stats(h)  # drop the counts of the frames above
        abi.set_traversal_mode(h, 7)
        abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), args.max_depth, sp, None, stream)
        torch.cuda.synchronize()
        full = abi.read_stats(h)
        abi.set_traversal_mode(h, 0)
        abi.kernel_times(h)
        ref_b = reference_bytes(full, W * H)
        extras["reference_equivalent"] = {
            "bytes_per_frame": ref_b, "GBps_at_this_frame_time": ref_b / (out["roofline"]["kernel_ms"] * 1e-3) / 1e9,
            "work_of_the_reference": {k: full[k] for k in ("box_tests", "node_visits", "tri_tests", "mt_tests")},
            "work_visited_by_this_kernel": {k: last_counters[k] for k in ("box_tests", "node_visits", "tri_tests", "mt_tests")},
            "note": "SURVEY 8(d) bytes of the reference's traversal of the golden camera's frame (no block / subtree "
                    "boxes, one box per ray); NOT this kernel's traffic -- shown for comparison only"}
        out["extras"] = extras
        if not args.no_cpu_baseline:
            chunk = (0, 0, W, H) if args.cpu_sample == "full" else (0, (H * 3) // 8, W, max(H // 4, 1))
            abi.render_chunk(h, sens, W, H, chunk=chunk, max_depth=args.max_depth)
            g = abi.render_chunk(h, sens, W, H, chunk=chunk, max_depth=args.max_depth)
            rc = sum(g["stats"][k] for k in RAY_KEYS)
            try:
                out["cpu_baseline"] = cpu_baseline(info["obj"], cam, lights, W, H, chunk, args.max_depth, rc)
                out["cpu_baseline"]["gpu_same_sample_Mray_s"] = rc / (g["stats"]["kernel_ms"] * 1e-3) / 1e6
                out["cpu_baseline"]["gpu_same_sample_note"] = ("one mt_render_chunk call (host buffers) of the golden camera's frame WITH the work "
                                                              "counters on (that is where the ray count comes from): kernels about a tenth slower "
                                                              "than the timed, counter-free ones -- not comparable with `value`")
            except Exception as e:  # the bench line must still come out
                out["cpu_baseline"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        flag = torch.tensor([1.0 if mismatch else 0.0], dtype=torch.float64, device=xdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        mismatch = bool(flag.item() > 0)
        dist.barrier()
        dist.destroy_process_group()
    if mismatch:
        if rank == 0:
            print("bench.py: PARITY MISMATCH (see the \"parity\" / \"extras\" fields)", file=sys.stderr)
        sys.exit(4)


if __name__ == "__main__":
    main()
