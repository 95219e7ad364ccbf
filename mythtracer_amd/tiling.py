"""Tile bookkeeping for multi-GPU frames.

Mirrors the master's work split in the reference: the frame is cut into
tile_w x tile_h WorkChunks in row-major order, edge tiles clipped
(VerStarting/main_net_master.cc:195-221 GenerateWork), each rendered with the
full-image sensor and blitted back by (chunk_y + j) * image_width + chunk_x + i
(main_net_master.cc:223-236 BlitWorkChunk).  Here tile k belongs to rank
k mod world_size, so that expensive image regions are spread over all GPUs.

The device-side equivalents are mt_render_tiles_device / mt_blit_tiles_device
of the C ABI; this module is the host-side description of the same layout (and
a numpy blit used by the CPU tests of the distributed path).
"""
from __future__ import annotations

import numpy as np


def tile_grid(image_w: int, image_h: int, tile_w: int, tile_h: int):
    """(tiles_x, tiles_y)."""
    return (image_w + tile_w - 1) // tile_w, (image_h + tile_h - 1) // tile_h


def tile_rect(k: int, image_w: int, image_h: int, tile_w: int, tile_h: int):
    """(x, y, w, h) of tile k, clipped to the image like GenerateWork does."""
    tx, _ = tile_grid(image_w, image_h, tile_w, tile_h)
    x0, y0 = (k % tx) * tile_w, (k // tx) * tile_h
    return x0, y0, min(tile_w, image_w - x0), min(tile_h, image_h - y0)


def rank_tiles(image_w: int, image_h: int, tile_w: int, tile_h: int, rank: int, world: int):
    """(first_tile, tile_stride, n_tiles) of one rank: tiles rank, rank+world, ..."""
    tx, ty = tile_grid(image_w, image_h, tile_w, tile_h)
    total = tx * ty
    n = 0 if rank >= total else (total - rank + world - 1) // world
    return rank, world, n


def max_tiles_per_rank(image_w, image_h, tile_w, tile_h, world):
    return rank_tiles(image_w, image_h, tile_w, tile_h, 0, world)[2]


def slot_bytes(tile_w: int, tile_h: int) -> int:
    return tile_w * tile_h * 3


def blit_tiles(image: np.ndarray, tiles: np.ndarray, tile_w: int, tile_h: int,
               first_tile: int, tile_stride: int, n_tiles: int) -> None:
    """BlitWorkChunk for a buffer of tile slots (numpy, host).  `tiles` is a
    flat uint8 array of n_tiles slots of tile_w*tile_h*3 bytes; slot j holds
    the chunk-local row-major bitmap of tile first_tile + j*tile_stride."""
    image_h, image_w, _ = image.shape
    sb = slot_bytes(tile_w, tile_h)
    flat = np.asarray(tiles, dtype=np.uint8).reshape(-1)
    for j in range(n_tiles):
        x0, y0, cw, ch = tile_rect(first_tile + j * tile_stride, image_w, image_h, tile_w, tile_h)
        chunk = flat[j * sb: j * sb + cw * ch * 3].reshape(ch, cw, 3)
        image[y0:y0 + ch, x0:x0 + cw] = chunk
