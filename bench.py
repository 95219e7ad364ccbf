#!/usr/bin/env python3
"""bench.py — the hot path on BASELINE.json's headline configuration.

One "step" = one frame through MythTracer::RayTrace's pixel loop on the GPU
(mt_render_chunk_device / mt_render_tiles_device of the C ABI): default
workload = configs[2] of BASELINE.json: the ~100k-triangle room (synthetic
stand-in for the unavailable living-room model, mythtracer_amd/scenegen.py),
1920x1080, 3 lights with shadow rays, the reference's MAX_RECURSION_LEVEL = 5.
Scene, lights and sensor are resident in HBM before the timed region; the
frame stays in HBM (the PCIe-inclusive rate is in DESIGN.md).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, every rank holds a scene replica (as every worker
does in the reference, main_net_worker.cc:29-32), renders the 64x64 tiles
k = rank (mod N) of ONE frame, and the tile buffers are gathered to rank 0 over
RCCL and blitted into the frame (main_net_master.cc:223-236) — the reference's
only exchange step.  Pixels are independent units, so the default is WEAK
scaling: the frame grows with N at the same camera and aspect (16k x 9k pixels,
k = round(120 sqrt(N)): 1920x1080, 2720x1530, 3840x2160 = BASELINE configs[4],
5440x3060), i.e. every GPU keeps about one 1080p frame's worth of pixels.
--scaling strong keeps the 1920x1080 frame for every N instead; that variant is
bounded by the longest per-pixel ray chain (about 8 ms of the 9.7 ms frame, see
DESIGN.md section 6), not by the GPUs.

Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(c: dict, pixels: int) -> int:
    """bytes the reference's algorithm touches per frame —
    48 B per node/child box test, 48 B per triangle pre-filter test, 72 B per
    Möller–Trumbore test, vertices+normals+material per shaded hit, 3 B/pixel."""
    return (48 * c["box_tests"] + 48 * c["tri_tests"] + 72 * c["mt_tests"] +
            (72 + 72 + 136) * c["shaded_hits"] + 3 * pixels)


def cpu_baseline(scene_obj, cam, lights, W, H, chunk, rays_in_chunk, max_depth):
    """Times the reference itself (oracle/_ref, built from /root/reference in
    the build container) on a bounded sample; falls back to our CPU port."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib
    cores = os.cpu_count() or 1
    sample = "chunk x=%d y=%d %dx%d of the %dx%d frame, all host threads" % (*chunk, W, H)
    if orclib.have_ref() and max_depth == 5:  # the reference's depth is a compile-time 5
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
    rays = sum(r["counters"][k] for k in ("rays_primary", "rays_secondary", "rays_shadow"))
    return {"value": rays / r["seconds"] / 1e6, "unit": "Mray/s", "cores": cores,
            "kind": "port", "sample": sample, "seconds": r["seconds"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=0, help="default: 1920, or scaled with --gpus (weak scaling)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--max-depth", type=int, default=5)
    ap.add_argument("--scene", default="room")
    ap.add_argument("--tile", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    if world > 1:
        if emulate:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    xdev = torch.device("cpu") if emulate else dev  # where collective operands live

    if args.width > 0 and args.height > 0:
        W, H = args.width, args.height
    elif args.scaling == "weak":
        k = int(round(120.0 * (world ** 0.5)))
        W, H = 16 * k, 9 * k          # N=1: 1920x1080, N=4: 3840x2160
    else:
        W, H = 1920, 1080
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
    n_max = tiling.max_tiles_per_rank(W, H, tw, th, world)
    frame = torch.zeros((H, W, 3), dtype=torch.uint8, device=dev)
    if world > 1:
        mine = torch.zeros(n_max * tiling.slot_bytes(tw, th), dtype=torch.uint8, device=dev)
        gathered = ([torch.zeros_like(mine, device=xdev) for _ in range(world)] if rank == 0 else None)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

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

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    abi.read_stats(h)              # drop warm-up counts
    abi.kernel_times(h)            # ... and warm-up kernel durations
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    counters = abi.read_stats(h)   # this rank, all timed steps
    # per-kernel device durations of the timed steps: HIP events the library
    # records on the launch stream around each of its two kernels
    k_primary, k_render = abi.kernel_times(h, 64)

    keys = ["rays_primary", "rays_secondary", "rays_shadow", "box_tests", "node_visits",
            "tri_tests", "mt_tests", "shaded_hits", "wave_node_steps", "wave_tri_steps"]
    vec = torch.tensor([float(counters[k]) for k in keys], dtype=torch.float64, device=xdev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    kmax = torch.tensor([float(k_render.mean()) if len(k_render) else 0.0,
                         float(k_primary.mean()) if len(k_primary) else 0.0,
                         sum(kernel_ms) / max(len(kernel_ms), 1)], dtype=torch.float64, device=xdev)
    if world > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    tot = {k: int(v) for k, v in zip(keys, vec.tolist())}
    per_frame = {k: v // max(args.steps, 1) for k, v in tot.items()}
    rays = tot["rays_primary"] + tot["rays_secondary"] + tot["rays_shadow"]

    if rank == 0:
        img = frame.cpu().numpy()
        sha = hashlib.sha256(img.tobytes()).hexdigest()
        sha_cmp, parity_ok = sha, "frame identical to the reference's"
        golden = None
        gpath = os.path.join(ROOT, "tests", "golden", "frames.json")
        if os.path.exists(gpath):
            key = "%s_%dx%d_d%d" % (args.scene, W, H, args.max_depth)
            golden = json.load(open(gpath)).get(key, {}).get("sha256")
            if golden is None and (W, H) == (3840, 2160):
                # Sensor::GetRay divides the same corner vectors by W and H: the ray of
                # 4K pixel (2x, 2y) is bit-for-bit that of 1080p pixel (x, y), so the even
                # pixels of this frame must be the reference's 1080p frame
                golden = json.load(open(gpath)).get("%s_1920x1080_d%d" % (args.scene, args.max_depth), {}).get("sha256")
                sha_cmp = hashlib.sha256(np.ascontiguousarray(img[::2, ::2]).tobytes()).hexdigest()
                parity_ok = "even pixels identical to the reference's 1920x1080 frame"
        # A step is two launches: mt::primary_kernel (primary rays, block cost
        # classes) and mt::render_kernel (shading, shadow + secondary rays) —
        # the dominant one.  Per-launch figures of THIS rank at N=1; at N>1 the
        # slowest rank's durations.
        k_ms, k_primary_ms, k_step_ms = (float(x) for x in kmax.tolist())
        my = counters if world == 1 else None
        roof = None
        if world == 1:
            pf = {k: my[k] // max(args.steps, 1) for k in keys}
            # Algorithmic work = the reference's un-pruned traversal:
            # counted by untimed frames in traversal mode 7, which visits every
            # subtree the reference visits (the timed frames skip those a ray
            # provably cannot hit; the image is the same).  A second one with no
            # lights and recursion 0 traces the primary rays only and gives
            # primary_kernel's share.
            scratch = torch.zeros_like(frame)
            evaluated = {k: pf[k] for k in ("box_tests", "node_visits", "tri_tests", "mt_tests")}
            abi.set_traversal_mode(h, 7)
            abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), args.max_depth,
                                    ctypes.c_void_p(scratch.data_ptr()), None, stream)
            torch.cuda.synchronize()
            full = abi.read_stats(h)
            reference_evaluates = {k: full[k] for k in ("box_tests", "node_visits", "tri_tests", "mt_tests")}
            abi.set_lights(h, [])
            abi.render_chunk_device(h, sens, W, H, (0, 0, W, H), 0,
                                    ctypes.c_void_p(scratch.data_ptr()), None, stream)
            torch.cuda.synchronize()
            prim = abi.read_stats(h)
            abi.kernel_times(h)
            abi.set_lights(h, lights)
            abi.set_traversal_mode(h, 0)
            pf = {k: full[k] for k in keys}
            # With cost history (every timed step of a default run) render_kernel
            # traces the primary rays too and the first kernel of the step is the
            # 50-us schedule_kernel; without it they belong to primary_kernel.
            primary_inside = k_primary_ms * 20.0 < k_ms
            if not primary_inside:
                for k in ("box_tests", "tri_tests", "mt_tests"):
                    pf[k] -= prim[k]
            alg = algorithmic_bytes(pf, W * H)
            alg_primary = 48 * prim["box_tests"] + 48 * prim["tri_tests"] + 72 * prim["mt_tests"] + 12 * W * H
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tpath):
                traffic = json.load(open(tpath)).get("%s_%dx%d_d%d" % (args.scene, W, H, args.max_depth))
            ach = alg / (k_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                    "kernel": "mt::render_kernel", "kernel_ms": k_ms,
                    "algorithmic_bytes_per_launch": alg,
                    "launches_averaged": int(len(k_render)),
                    "primary_rays_traced_by_this_kernel": bool(primary_inside),
                    "other_kernels": {"mt::schedule_kernel (mt::primary_kernel in a launch without cost history)": {
                        "kernel_ms": k_primary_ms},
                        "primary_rays_algorithmic_bytes": alg_primary},
                    "step_ms_device": k_step_ms,
                    "note": "algorithmic bytes = SURVEY 8(d) bytes of the reference's un-pruned "
                            "traversal (counted by an untimed frame in traversal mode 7); the kernel "
                            "serves one box to 64 rays with a scalar load and rules out most blocks and "
                            "subtrees by 24-byte union boxes, so it is latency/VALU bound, not HBM bound "
                            "(DESIGN.md section 5); traffic = memory-side bytes per launch from the "
                            "rocprofv3 PMC passes under profiles/",
                    # both normalisations -- the bytes of the nodes this kernel
                    # actually visits (subtrees it can rule out are not counted) ...
                    "achieved_own_counters": (48 * evaluated["box_tests"] + 48 * evaluated["tri_tests"]
                                              + 72 * evaluated["mt_tests"]) / (k_step_ms * 1e-3) / 1e9,
                    "frame_work_visited_by_the_kernels": evaluated,
                    "frame_work_of_the_reference": reference_evaluates}
        out = {
            "metric": "Mray/s (primary+shadow+secondary; ray = one OctTree::IntersectRay)",
            "value": rays / elapsed / 1e6, "unit": "Mray/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling if not (args.width > 0 and args.height > 0) else "strong",
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s scene (%d triangles, synthetic stand-in for the living-room "
                                   ".obj), %dx%d, %d lights with shadow rays, max recursion %d"
                                   % (args.scene, info["triangles"], W, H, len(lights), args.max_depth),
                       "tile": "%dx%d interleaved over %d rank(s)" % (tw, th, world) if world > 1
                               else "whole frame per launch, 8x8-pixel work items",
                       "scene_sha256": info["sha256"]},
            "frame_ms_wall": elapsed / max(args.steps, 1) * 1e3,
            "rays_per_frame": {k: per_frame[k] for k in ("rays_primary", "rays_secondary", "rays_shadow")},
            "Mray_s_primary_plus_shadow": (tot["rays_primary"] + tot["rays_shadow"]) / elapsed / 1e6,
            "frame_sha256": sha,
            "parity": (None if golden is None else (parity_ok if golden == sha_cmp else "MISMATCH vs reference")),
            "scene_load_s": t_load,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            chunk = (0, (H * 3) // 8, W, max(H // 4, 1))  # middle band, a quarter of the frame
            g = abi.render_chunk(h, sens, W, H, chunk=chunk, max_depth=args.max_depth)
            rc = sum(g["stats"][k] for k in ("rays_primary", "rays_secondary", "rays_shadow"))
            try:
                out["cpu_baseline"] = cpu_baseline(info["obj"], cam, lights, W, H, chunk, rc, args.max_depth)
                out["cpu_baseline"]["gpu_same_sample_Mray_s"] = rc / (g["stats"]["kernel_ms"] * 1e-3) / 1e6
            except Exception as e:  # the bench line must still come out
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
