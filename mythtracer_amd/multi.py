"""The multi-GPU exchange step: gather every rank's tile buffer on rank 0 and
blit the tiles into the frame — what the reference's master does with PXLS
packets over TCP (VerStarting/main_net_master.cc:131-160, 223-236), here one
`torch.distributed` gather (RCCL over xGMI on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from . import tiling


def gather_and_blit(dist, mine, gathered, rank, world, image_w, image_h, tile_w, tile_h, blit):
    """mine: this rank's tile slots (equal-sized tensor on every rank);
    gathered: list of `world` such tensors on rank 0, else None;
    blit(slots, first_tile, tile_stride, n_tiles): writes tiles into the frame."""
    dist.gather(mine, gathered, dst=0)
    if rank == 0:
        for r in range(world):
            first, stride, n = tiling.rank_tiles(image_w, image_h, tile_w, tile_h, r, world)
            blit(gathered[r], first, stride, n)


def gather_and_blit_lists(dist, mine, gathered, rank, world, list_of, blit):
    """The same exchange for cost-balanced ownership (tiling.deal_tiles / mt_deal_tiles_device): every rank holds
    the tiles of ITS list, slot j = tile list[j]; list_of(r) gives rank r's list on rank 0 (every rank can compute
    all of them: the order is a function of the all-reduced cost map); blit(slots, tile_list) writes them."""
    dist.gather(mine, gathered, dst=0)
    if rank == 0:
        for r in range(world):
            blit(gathered[r], list_of(r))
