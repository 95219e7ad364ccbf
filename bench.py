#!/usr/bin/env python3
"""bench.py — the hot path on BASELINE.json's headline configurations.

One "step" = one frame through MythTracer::RayTrace's pixel loop on the GPU
(mt_render_chunk_device / mt_render_tiles_device of the C ABI).  Scene, lights
and sensor are resident in HBM before the timed region; the frame stays in HBM
(the PCIe-inclusive rate is in DESIGN.md).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workloads (the ~100k-triangle room of mythtracer_amd/scenegen.py stands in for
the unavailable living-room model; 3 lights with shadow rays, the reference's
MAX_RECURSION_LEVEL = 5):

  N = 1   BASELINE configs[2]: 1920x1080, one launch for the whole frame.
  N > 1   BASELINE configs[4]: 3840x2160, STRONG scaling: every rank holds a
          scene replica (as every worker does in the reference,
          main_net_worker.cc:29-32), renders the 64x64 tiles k = rank (mod N) of
          the ONE frame, and the tile buffers are gathered to rank 0 over RCCL
          and blitted into the frame (main_net_master.cc:223-236) -- the
          reference's only exchange step.  `--scaling weak` grows the frame with
          N instead (16k x 9k pixels, k = round(120 sqrt(N))).
  --width/--height/--max-depth/--scene select the other BASELINE configurations.

Rank 0 prints ONE JSON line; the process exits non-zero if the frame does not
match the golden frame of the reference.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import math
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
RAY_KEYS = ("rays_primary", "rays_secondary", "rays_shadow")


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--width", type=int, default=0, help="default: 1920 at N = 1, 3840 at N > 1")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--scene", default="room")
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", choices=("full", "band"), default="full",
                    help="reference timed on the whole frame (about 30 s) or on a quarter-frame band")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extras (cold frame, moving camera, reference-work counts)")
    ap.add_argument("--moving-frames", type=int, default=12)
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import mythtracer_amd as M
    from mythtracer_amd import multi, scenegen, tiling
    from mythtracer_amd.binding import sensor as host_sensor

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run"
                  % (args.gpus, world), file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
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
    if world > 1:  # before anything else touches the device
        if emulate:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
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
    t_load = time.time() - t_load
    sens = host_sensor(cam, W, H)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    tw = th = args.tile
    first, stride, n_mine = tiling.rank_tiles(W, H, tw, th, rank, world)
    n_max = tiling.max_tiles_per_rank(W, H, tw, th, world)   # equal slot counts on every rank
    frame = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    if world > 1:
        mine = torch.zeros(n_max * tiling.slot_bytes(tw, th), dtype=torch.uint8, device=dev)
        gathered = ([torch.zeros_like(mine, device=xdev) for _ in range(world)] if rank == 0 else None)

    def E():
        return torch.cuda.Event(enable_timing=True)
    ev = [(E(), E(), E()) for _ in range(args.steps)]  # before render, after render, after gather + blit

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        if world == 1:
            abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), args.max_depth,
                                    ctypes.c_void_p(frame.data_ptr()), None, stream)
        else:
            abi.render_tiles_device(h, sens, W, H, tw, th, first, stride, n_mine, args.max_depth,
                                    ctypes.c_void_p(mine.data_ptr()), stream)
        if i is not None:
            ev[i][1].record()
        if world > 1:
            def blit(slots, f_r, s_r, n_r):
                slots = slots.to(dev)  # no-op except in the rehearsal mode
                abi.blit_tiles_device(h, W, H, tw, th, f_r, s_r, n_r, ctypes.c_void_p(slots.data_ptr()),
                                      ctypes.c_void_p(frame.data_ptr()), stream)
                if emulate:
                    torch.cuda.synchronize()  # `slots` is a temporary here
            multi.gather_and_blit(dist, mine.to(xdev), gathered, rank, world, W, H, tw, th, blit)
        if i is not None:
            ev[i][2].record()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    abi.read_stats(h)              # drop warm-up counts
    # One untimed frame WITH the work counters gives the frame's ray counts and
    # requested bytes (they are the same for every step: the frame is the same);
    # the timed steps then run the kernels built without the counters
    # (mt_scene_set_stats: about 7 % of a frame goes into counting).
    step()
    fence()
    counters = abi.read_stats(h)   # this rank, one frame
    abi.set_stats(h, False)
    step()                         # first launch of the counter-free kernels
    fence()
    abi.kernel_times(h)            # drop the durations so far
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    # per-kernel device durations of the timed steps: HIP events the library
    # records on the launch stream around each of its kernels (at most the last 64)
    k_order, k_frame = abi.kernel_times(h, 64)
    abi.set_stats(h, True)
    # the same, counters on, for comparison (16 steps)
    fence()
    t1 = time.perf_counter()
    for _ in range(16):
        step()
    fence()
    elapsed_counting = (time.perf_counter() - t1) / 16.0
    abi.read_stats(h)
    abi.kernel_times(h)

    keys = ["rays_primary", "rays_secondary", "rays_shadow", "box_tests", "node_visits",
            "tri_tests", "mt_tests", "shaded_hits", "wave_node_steps", "wave_tri_steps",
            "bytes_scalar", "bytes_vector"]
    vec = torch.tensor([float(counters[k]) for k in keys], dtype=torch.float64, device=xdev)
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
    per_frame = {k: int(v) for k, v in zip(keys, vec.tolist())}   # all ranks, one frame
    tot = {k: v * args.steps for k, v in per_frame.items()}
    rays = tot["rays_primary"] + tot["rays_secondary"] + tot["rays_shadow"]
    mismatch = False

    if rank == 0:
        img = frame.cpu().numpy()
        sha = hashlib.sha256(img.tobytes()).hexdigest()
        sha_cmp, parity_ok = sha, "frame identical to the reference's"
        golden, golden_from = None, None
        gpath = os.path.join(ROOT, "tests", "golden", "frames.json")
        if os.path.exists(gpath):
            frames = json.load(open(gpath))
            key = "%s_%dx%d_d%d" % (args.scene, W, H, args.max_depth)
            golden = frames.get(key, {}).get("sha256")
            golden_from = frames.get(key, {}).get("made_by", "the compiled reference")
            if golden is None and (W, H) == (3840, 2160):
                # Sensor::GetRay divides the same corner vectors by W and H: the ray of
                # 4K pixel (2x, 2y) is bit-for-bit that of 1080p pixel (x, y), so the even
                # pixels of this frame must be the reference's 1080p frame
                k2 = "%s_1920x1080_d%d" % (args.scene, args.max_depth)
                golden = frames.get(k2, {}).get("sha256")
                golden_from = frames.get(k2, {}).get("made_by", "the compiled reference")
                sha_cmp = hashlib.sha256(np.ascontiguousarray(img[::2, ::2]).tobytes()).hexdigest()
                parity_ok = "even pixels identical to the 1920x1080 golden frame"
            if golden is not None and golden_from != "the compiled reference":
                parity_ok += " (golden made by: %s)" % golden_from
        mismatch = golden is not None and golden != sha_cmp
        k_ms, k_order_ms, k_step_ms, k_exchange_ms = (float(x) for x in kmax.tolist())

        # ---- roofline of the dominant kernel (the frame kernel).  Numerator: the
        # bytes the kernel REQUESTS per launch, counted by the kernel itself at its
        # load/store sites (mt_stats.bytes_scalar + bytes_vector; DESIGN.md section 5
        # lists them) -- the work this kernel does, not the reference's.
        req = per_frame["bytes_scalar"] + per_frame["bytes_vector"]
        if world > 1:
            req = None  # summed over ranks: not a per-launch figure
        roof = None
        if world == 1:
            ach = req / (k_ms * 1e-3) / 1e9
            wl_key = "%s_%dx%d_d%d" % (args.scene, W, H, args.max_depth)
            traffic, pmc = None, None
            tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tpath):
                t = json.load(open(tpath)).get(wl_key)
                if isinstance(t, dict):
                    traffic, pmc = t.get("hbm_bytes_per_launch"), t
                else:
                    traffic = t
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                    "kernel": "mt::render_kernel (throughput engine) / mt::pool_kernel (latency engine)",
                    "kernel_ms": k_ms, "launches_averaged": int(len(k_frame)),
                    "requested_bytes_per_launch": {"scalar_wave_uniform": per_frame["bytes_scalar"],
                                                   "vector_per_lane": per_frame["bytes_vector"]},
                    "other_kernels_ms": {"work order (schedule / probe / primary)": k_order_ms},
                    "step_ms_device": k_step_ms,
                    "note": "achieved = bytes requested by the frame kernel per launch (its own counters) / its "
                            "average duration; most are served by the scalar cache, L1 and L2 (working set 33 MB), "
                            "so the HBM fraction is low by construction: the kernel is bound by VALU issue and "
                            "dependent-load latency, see binding_resource and DESIGN.md section 5"}
            if pmc:
                roof["binding_resource"] = {k: pmc[k] for k in pmc if k != "hbm_bytes_per_launch"}

        out = {
            "metric": "Mray/s (primary+shadow+secondary; ray = one OctTree::IntersectRay)",
            "value": rays / elapsed / 1e6, "unit": "Mray/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s scene (%d triangles, synthetic stand-in for the living-room "
                                   ".obj), %dx%d, %d lights with shadow rays, max recursion %d%s"
                                   % (args.scene, info["triangles"], W, H, len(lights), args.max_depth,
                                      " = BASELINE configs[2]" if (W, H, args.max_depth, world) == (1920, 1080, 5, 1)
                                      else (" = BASELINE configs[4]" if (W, H, args.max_depth) == (3840, 2160, 5) and world > 1
                                            else "")),
                       "tile": "%dx%d tiles interleaved over %d ranks, gathered to rank 0 (RCCL) and blitted" % (tw, th, world)
                               if world > 1 else "whole frame per launch, 8x8-pixel work items",
                       "regime": "warm: every timed step re-renders the same frame and is scheduled from the block "
                                 "costs measured in the previous step (cold_frame_ms / moving_camera below give "
                                 "the other regimes)",
                       "scene_sha256": info["sha256"]},
            "frame_ms_wall": elapsed / max(args.steps, 1) * 1e3,
            "frame_ms_wall_with_work_counters": elapsed_counting * 1e3,
            "work_counters": "off in the timed steps (mt_scene_set_stats(scene, 0)); ray counts and requested bytes "
                             "are those of one untimed frame of the same workload rendered with the counters on",
            "render_ms_device": k_step_ms,
            "rays_per_frame": {k: per_frame[k] for k in RAY_KEYS},
            "Mray_s_primary_plus_shadow": (tot["rays_primary"] + tot["rays_shadow"]) / elapsed / 1e6,
            "frame_sha256": sha,
            "parity": (None if golden is None else ("MISMATCH vs golden frame" if mismatch else parity_ok)),
            "scene_load_s": t_load,
            "roofline": roof,
        }
        if world > 1:
            out["exchange_ms_device"] = {"gather_plus_blit_rank0": k_exchange_ms,
                                         "bytes_gathered": int(n_max * tiling.slot_bytes(tw, th) * world)}

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

        # (1) cold frame: no cost history (first frame of a geometry).  Like the timed steps, (1) and (2) run the
        # kernels built without the work counters.
        abi.set_stats(h, False)
        abi.set_scheduling(h, True)  # forgets the recorded costs
        extras["cold_frame_ms"] = timed_frame(sens)
        # (2) moving camera: the reference's loop turns the camera 2 degrees per frame
        # (main_local.cc:51-76); every frame is scheduled from the PREVIOUS frame's costs.
        # Each frame is checked on a crop against the oracle.
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orclib
        orc = orclib.OracleScene(info["obj"])
        orc.set_lights(lights)
        mv, crops_ok = [], True
        for f in range(args.moving_frames):
            cam_f = list(cam)
            cam_f[4] = cam[4] + 2.0 * (f + 1)  # yaw
            s_f = host_sensor(cam_f, W, H)
            mv.append(timed_frame(s_f))
            cw, ch = 96, 48
            cx, cy = (211 * f) % (W - cw), (H // 3 + 37 * f) % (H - ch)
            want = orc.render(cam_f, W, H, chunk=(cx, cy, cw, ch), max_level=args.max_depth)["rgb"]
            got = scratch[cy:cy + ch, cx:cx + cw].cpu().numpy()
            crops_ok = crops_ok and bool(np.array_equal(got, want))
        abi.set_stats(h, True)
        extras["work_counters"] = "off for cold_frame_ms and moving_camera, as in the timed steps"
        extras["moving_camera"] = {"frames": args.moving_frames, "yaw_step_deg": 2.0,
                                   "ms_mean": sum(mv) / len(mv), "ms_max": max(mv), "ms_min": min(mv),
                                   "every_frame_crop_equals_oracle": crops_ok}
        mismatch = mismatch or not crops_ok
        # (3) the work of the REFERENCE's un-pruned traversal on this frame (traversal
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
            "work_visited_by_this_kernel": {k: per_frame[k] for k in ("box_tests", "node_visits", "tri_tests", "mt_tests")},
            "note": "SURVEY 8(d) bytes of the reference's traversal (no block / subtree boxes, one box per ray); "
                    "NOT this kernel's traffic -- shown for comparison only"}
        out["extras"] = extras
        if not args.no_cpu_baseline:
            chunk = (0, 0, W, H) if args.cpu_sample == "full" else (0, (H * 3) // 8, W, max(H // 4, 1))
            abi.render_chunk(h, sens, W, H, chunk=chunk, max_depth=args.max_depth)  # (the camera has just jumped back: no usable history)
            g = abi.render_chunk(h, sens, W, H, chunk=chunk, max_depth=args.max_depth)
            rc = sum(g["stats"][k] for k in RAY_KEYS)
            try:
                out["cpu_baseline"] = cpu_baseline(info["obj"], cam, lights, W, H, chunk, args.max_depth, rc)
                out["cpu_baseline"]["gpu_same_sample_Mray_s"] = rc / (g["stats"]["kernel_ms"] * 1e-3) / 1e6
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
