"""One rank of the CPU rehearsal of the multi-GPU frame path (gloo).  The
renderer is stood in by the CPU oracle; everything else — tile ownership,
slot layout, gather, blit — is the code bench.py runs on GPUs."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orclib  # noqa: E402
from mythtracer_amd import multi, tiling  # noqa: E402


def main():
    obj, out_path, W, H, T = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    cam = (50, 50, -120, 0, 0, 0, 60)
    lights = [(50, 90, 50, .3, .3, .3, 1, 1, 1, 1, 1, 1)]
    o = orclib.OracleScene(obj)
    o.set_lights(lights)
    first, stride, n = tiling.rank_tiles(W, H, T, T, rank, world)
    n_max = tiling.max_tiles_per_rank(W, H, T, T, world)
    sb = tiling.slot_bytes(T, T)
    mine = torch.zeros(n_max * sb, dtype=torch.uint8)
    for j in range(n):
        x, y, cw, ch = tiling.tile_rect(first + j * stride, W, H, T, T)
        rgb = o.render(cam, W, H, chunk=(x, y, cw, ch), nthreads=1)["rgb"]
        mine[j * sb: j * sb + cw * ch * 3] = torch.from_numpy(rgb.reshape(-1).copy())
    frame = np.zeros((H, W, 3), dtype=np.uint8)
    gathered = [torch.zeros_like(mine) for _ in range(world)] if rank == 0 else None
    multi.gather_and_blit(dist, mine, gathered, rank, world, W, H, T, T,
                          lambda slots, f, s, k: tiling.blit_tiles(frame, slots.numpy(), T, T, f, s, k))
    dist.barrier()
    if rank == 0:
        np.save(out_path, frame)
    if len(sys.argv) > 6 and sys.argv[6] == "balanced":
        # Second frame with COST-BALANCED ownership, as bench.py --gpus N does it: every rank writes a cost per 8x8
        # block of its tiles into a frame-wide map (here: the block's summed pixel values, standing in for measured
        # wave cycles), the maps are all-reduced (MAX), every rank orders the tiles by that map and takes its deal.
        mw, mh = (W + 7) // 8, (H + 7) // 8
        cmap = torch.zeros((mh, mw), dtype=torch.int32)
        for j in range(n):
            x, y, cw, ch = tiling.tile_rect(first + j * stride, W, H, T, T)
            rgb = mine[j * sb: j * sb + cw * ch * 3].numpy().reshape(ch, cw, 3).astype(np.int64)
            for by in range(0, ch, 8):
                for bx in range(0, cw, 8):
                    cmap[(y + by) // 8, (x + bx) // 8] = 1 + int(rgb[by:by + 8, bx:bx + 8].sum())
        dist.all_reduce(cmap, op=dist.ReduceOp.MAX)
        tx, ty = tiling.tile_grid(W, H, T, T)
        order = tiling.order_tiles(cmap.numpy().astype(np.uint32), W, H, T, T)
        lst = tiling.deal_tiles(order, tx * ty, world, rank)
        assert len(lst) <= n_max
        mine2 = torch.zeros(n_max * sb, dtype=torch.uint8)
        for j, t in enumerate(lst):
            x, y, cw, ch = tiling.tile_rect(int(t), W, H, T, T)
            rgb = o.render(cam, W, H, chunk=(x, y, cw, ch), nthreads=1)["rgb"]
            mine2[j * sb: j * sb + cw * ch * 3] = torch.from_numpy(rgb.reshape(-1).copy())
        frame2 = np.zeros((H, W, 3), dtype=np.uint8)
        gathered2 = [torch.zeros_like(mine2) for _ in range(world)] if rank == 0 else None
        multi.gather_and_blit_lists(dist, mine2, gathered2, rank, world,
                                    lambda r: tiling.deal_tiles(order, tx * ty, world, r),
                                    lambda slots, tl: tiling.blit_tile_list(frame2, slots.numpy(), T, T, tl))
        dist.barrier()
        if rank == 0:
            np.save(out_path + ".balanced.npy", frame2)
            np.save(out_path + ".order.npy", order)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
