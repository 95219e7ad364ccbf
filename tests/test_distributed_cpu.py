"""world_size-2 (and 3) rehearsal of the N>1 frame path on CPU with gloo."""
import os
import subprocess
import sys

import numpy as np
import pytest

import orclib
from conftest import CORNELL, ROOT


@pytest.mark.parametrize("world,size,tile", [(2, (72, 40), 16), (3, (50, 37), 16)])
def test_tiles_gather_to_the_single_process_frame(world, size, tile, tmp_path):
    W, H = size
    out = str(tmp_path / "frame.npy")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + world),
           os.path.join(ROOT, "tests", "dist_worker.py"), CORNELL, out, str(W), str(H), str(tile), "balanced"]
    subprocess.run(cmd, check=True, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    got = np.load(out)
    o = orclib.OracleScene(CORNELL)
    o.set_lights([(50, 90, 50, .3, .3, .3, 1, 1, 1, 1, 1, 1)])
    want = o.render((50, 50, -120, 0, 0, 0, 60), W, H)["rgb"]
    assert np.array_equal(got, want)
    # ... and the second frame, whose tiles were dealt out by the all-reduced cost map (non-modular ownership)
    assert np.array_equal(np.load(out + ".balanced.npy"), want)
    order = np.load(out + ".order.npy")
    assert sorted(order.tolist()) == list(range(len(order))) and order.tolist() != list(range(len(order)))


def test_dealing_tiles_by_cost():
    """tiling.order_tiles / deal_tiles (the numpy restatement of mt_order_tiles_device / mt_deal_tiles_device that the GPU
    tests compare the kernels with): every tile exactly once, equal counts up to one, rounds of alternating direction,
    and cost sums closer together than the modular assignment's on a map with a few expensive tiles."""
    from mythtracer_amd import tiling
    W, H, T, world = 640, 360, 32, 8
    tx, ty = tiling.tile_grid(W, H, T, T)
    rnd = np.random.RandomState(5)
    cmap = rnd.randint(100, 200, size=((H + 7) // 8, (W + 7) // 8)).astype(np.uint32)
    cmap[10:22, 30:44] *= 40  # a glass sphere's worth of expensive blocks
    order = tiling.order_tiles(cmap, W, H, T, T)
    lists = [tiling.deal_tiles(order, tx * ty, world, r) for r in range(world)]
    assert sorted(np.concatenate(lists).tolist()) == list(range(tx * ty))
    assert max(len(l) for l in lists) - min(len(l) for l in lists) <= 1
    assert [tiling.dealt_tile_count(tx * ty, world, r) for r in range(world)] == [len(l) for l in lists]
    cost = np.zeros(tx * ty)
    for t in range(tx * ty):
        x, y, cw, ch = tiling.tile_rect(t, W, H, T, T)
        cost[t] = cmap[y // 8:(y + ch + 7) // 8, x // 8:(x + cw + 7) // 8].sum()
    assert (np.diff(cost[order]) <= 0).all()
    dealt = np.array([cost[l].sum() for l in lists])
    modular = np.array([cost[r::world].sum() for r in range(world)])
    assert dealt.max() / dealt.mean() < modular.max() / modular.mean()
    # no order yet: by tile number, still in rounds of alternating direction
    assert tiling.deal_tiles(None, 20, 8, 0).tolist() == [0, 15, 16] and tiling.deal_tiles(None, 20, 8, 7).tolist() == [7, 8]


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment (how a
    driver may start the N-GPU run): the parent -- before it imports torch or
    touches HIP -- starts torch.distributed.run with 2 ranks as a child process
    and relays their output and exit code.  There is no GPU here, so every rank
    ends with "no GPU visible" (exit 3): what is checked is that two ranks were
    started and that their failure reaches the caller -- not rc 2 from the parent."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, timeout=600, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr
    assert r.stderr.count("no GPU visible") == 2, r.stderr[-2000:]
    assert r.returncode not in (0, 2)
