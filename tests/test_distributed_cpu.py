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
           os.path.join(ROOT, "tests", "dist_worker.py"), CORNELL, out, str(W), str(H), str(tile)]
    subprocess.run(cmd, check=True, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    got = np.load(out)
    o = orclib.OracleScene(CORNELL)
    o.set_lights([(50, 90, 50, .3, .3, .3, 1, 1, 1, 1, 1, 1)])
    want = o.render((50, 50, -120, 0, 0, 0, 60), W, H)["rgb"]
    assert np.array_equal(got, want)
