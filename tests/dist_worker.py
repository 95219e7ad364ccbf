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
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
