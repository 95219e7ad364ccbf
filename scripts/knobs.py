"""The MT_DEBUG_* tuning variables of rounds 1-2 -> mt_scene_set_tuning.

The library reads no environment variable on its launch path any more
(include/mythtracer_hip.h, mt_scene_set_tuning); the experiment scripts keep their
old spelling -- they set os.environ["MT_DEBUG_..."] or receive it from a parent
sweep -- and call from_env(abi, h) afterwards.  (The DUMP facilities --
MT_DEBUG_ITEM_CYCLES, _PRINT_UNITS, _TIMELINE, _HEARTBEAT -- exist only in a
-DMT_DEBUG_KNOBS build, which reads them once in mt_scene_create:
MT_EXTRA_FLAGS=-DMT_DEBUG_KNOBS python -m mythtracer_amd.build --knobs.)"""
import os

_MAP = {"MT_DEBUG_BLEND": ("BLEND",), "MT_DEBUG_QUAD_SHARE": ("QUAD_SHARE",),
        "MT_DEBUG_QUAD_SHARE_MOVING": ("QUAD_SHARE_MOVING",), "MT_DEBUG_QUAD_KEEP": ("QUAD_KEEP",),
        "MT_DEBUG_QUAD_WORK": ("QUAD_WORK", "QUAD_WORK_MOVING"), "MT_DEBUG_CUT_SHARE": ("POOL_CUT_SHARE",),
        "MT_DEBUG_CELL_FACTOR": ("POOL_CELL_FACTOR",), "MT_DEBUG_POOL_BELOW": ("POOL_BELOW",),
        "MT_DEBUG_POOL_CAP": ("POOL_CAP",), "MT_DEBUG_BLOCKS_PER_CU": ("BLOCKS_PER_CU",),
        "MT_DEBUG_FORECAST_RADIUS": ("FORECAST_RADIUS",)}


def from_env(abi, h):
    for env, knobs in _MAP.items():
        if env in os.environ:
            for k in knobs:
                abi.set_tuning(h, k, float(os.environ[env]))
    for env, pair in (("MT_DEBUG_PIECE_TIME", ("POOL_PIECE_TIME1", "POOL_PIECE_TIME2")),
                      ("MT_DEBUG_PIECE_WORK", ("POOL_PIECE_WORK1", "POOL_PIECE_WORK2"))):
        if env in os.environ:
            a, b = (float(x) for x in os.environ[env].split(","))
            abi.set_tuning(h, pair[0], a)
            abi.set_tuning(h, pair[1], b)
    abi.set_tuning(h, "FORMS", 0.0 if os.environ.get("MT_DEBUG_NO_FORMS") else 1.0)
    if "MT_ENGINE" in os.environ:
        abi.set_engine(h, int(os.environ["MT_ENGINE"]))
