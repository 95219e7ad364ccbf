import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
CORNELL = os.path.join(ROOT, "tests", "scenes", "cornell_n.obj")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # torch ships its own copy of the HIP runtime.  If libmythtracer_hip.so pulls
    # in /opt/rocm's first, torch.cuda later reports "No HIP GPUs": let torch
    # initialise the runtime first whenever it is going to be used at all
    # (tests only; the product libraries do not depend on torch).
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass


@pytest.fixture(scope="session")
def native_libs():
    """Builds (if stale) and returns the paths of the product libraries."""
    from mythtracer_amd import build
    return build.build_all()


@pytest.fixture(scope="session")
def scenes(tmp_path_factory):
    """Generates the synthetic scenes once per session; checks that the text is
    the one the golden vectors were made from."""
    from mythtracer_amd import scenegen
    d = str(tmp_path_factory.mktemp("scenes"))
    want = json.load(open(os.path.join(GOLDEN, "scene_hashes.json")))
    out = {"cornell": CORNELL, "f2_decal": os.path.join(ROOT, "tests", "scenes", "f2_decal.obj")}
    for name in ("mini", "mini_nomtl", "room", "room_nomtl", "room_tex", "loft", "loft_fine"):
        info = scenegen.write_scene(name, d)
        if name in want:
            assert info["sha256"] == want[name], "scene generator drifted for " + name
        out[name] = info["obj"]
    return out


@pytest.fixture(scope="session")
def quirk_dir(tmp_path_factory):
    import quirk_files
    return quirk_files.write_all(str(tmp_path_factory.mktemp("quirks")))
