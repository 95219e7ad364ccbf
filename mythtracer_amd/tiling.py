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


def dealt_position(q: int, world: int, rank: int) -> int:
    """Position, in the cost order of the tiles, of the tile rank `rank` holds in round q: the rounds change
    direction (0 1 .. N-1, N-1 .. 1 0, ...) -- mt_order_tiles_device / deal_tiles_kernel of the C ABI."""
    return q * world + ((world - 1 - rank) if (q & 1) else rank)


def dealt_tile_count(n_tiles: int, world: int, rank: int) -> int:
    n = 0
    for q in (n_tiles // world - 1, n_tiles // world):
        if q >= 0 and dealt_position(q, world, rank) < n_tiles:
            n = q + 1
    return n


def order_tiles(cost_map: np.ndarray, image_w: int, image_h: int, tile_w: int, tile_h: int) -> np.ndarray:
    """Tiles by summed block cost, most expensive first, ties by tile number (numpy restatement of
    tile_cost_kernel + tile_order_kernel; cost_map = uint32 [map_h][map_w], one word per 8x8 block)."""
    tx, ty = tile_grid(image_w, image_h, tile_w, tile_h)
    cost = np.zeros(tx * ty, dtype=np.uint64)
    for t in range(tx * ty):
        x0, y0, cw, ch = tile_rect(t, image_w, image_h, tile_w, tile_h)
        cost[t] = cost_map[y0 >> 3:((y0 + ch - 1) >> 3) + 1, x0 >> 3:((x0 + cw - 1) >> 3) + 1].astype(np.uint64).sum()
    return np.lexsort((np.arange(tx * ty), -cost.astype(np.int64))).astype(np.int32)


def deal_tiles(order, n_tiles: int, world: int, rank: int) -> np.ndarray:
    """The rank's tile list in slot order (order None: by tile number)."""
    n = dealt_tile_count(n_tiles, world, rank)
    pos = np.array([dealt_position(q, world, rank) for q in range(n)], dtype=np.int64)
    return (pos if order is None else np.asarray(order)[pos]).astype(np.int32)


def blit_tile_list(image: np.ndarray, tiles: np.ndarray, tile_w: int, tile_h: int, tile_list) -> None:
    """BlitWorkChunk for a buffer whose slot j holds tile tile_list[j] (numpy, host)."""
    image_h, image_w, _ = image.shape
    sb = slot_bytes(tile_w, tile_h)
    flat = np.asarray(tiles, dtype=np.uint8).reshape(-1)
    for j, t in enumerate(tile_list):
        x0, y0, cw, ch = tile_rect(int(t), image_w, image_h, tile_w, tile_h)
        image[y0:y0 + ch, x0:x0 + cw] = flat[j * sb: j * sb + cw * ch * 3].reshape(ch, cw, 3)


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
