"""Builds the native libraries in-tree (mythtracer_amd/lib/):

  libmythtracer_hip.so   HIP kernels + C ABI (include/mythtracer_hip.h), gfx950
  libmythtracer_host.so  C++ facade (raytracer::MythTracer & co.) + ctypes shim

hipcc cross-compiles for gfx950 without a GPU, so this runs in the build
container; the .so files travel to the GPU box with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "lib")
CSRC = os.path.join(PKG, "csrc")
HOST = os.path.join(PKG, "host")
INC = os.path.join(ROOT, "include")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HIP_LIB = os.path.join(LIB, "libmythtracer_hip.so")
HOST_LIB = os.path.join(LIB, "libmythtracer_host.so")

# -ffp-contract=off: the kernels must round every product and sum separately,
# exactly like the reference built without FMA (VerStarting/Makefile:1-5).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17",
             "-fPIC", "-shared", "-Wall", "-Wno-pass-failed", "-Wno-unused-function",
             "-Wno-inline-asm"]  # (lds_dma16 declares M0 clobbered; clang notes that M0 is a reserved register)
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra",
              "-ffp-contract=off"]


def _newer(target: str, sources) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _files(d: str, exts):
    out = []
    for base, _, names in os.walk(d):
        out += [os.path.join(base, n) for n in names if n.endswith(exts)]
    return sorted(out)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIB, exist_ok=True)
    deps = _files(CSRC, (".hip", ".h")) + [os.path.join(INC, "mythtracer_hip.h")]
    if not force and _newer(HIP_LIB, deps):
        return HIP_LIB
    cmd = [HIPCC] + HIP_FLAGS + ["-I", INC, "-I", CSRC, "-o", HIP_LIB,
                                 os.path.join(CSRC, "mt_capi.hip")]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return HIP_LIB


def build_prof(verbose: bool = False) -> str:
    """Diagnostic variant with phase stamps (-DMT_PROF); never used by tests or bench."""
    out = os.path.join(LIB, "libmythtracer_hip_prof.so")
    cmd = [HIPCC] + HIP_FLAGS + ["-DMT_PROF"] + os.environ.get("MT_EXTRA_FLAGS", "").split() + ["-I", INC, "-I", CSRC, "-o", out,
                                 os.path.join(CSRC, "mt_capi.hip")]
    subprocess.check_call(cmd)
    return out


def build_diag(verbose: bool = False) -> str:
    """Diagnostic variant with per-unit step counters (-DMT_DIAG); never used by tests or bench."""
    out = os.path.join(LIB, "libmythtracer_hip_diag.so")
    cmd = [HIPCC] + HIP_FLAGS + ["-DMT_DIAG", "-I", INC, "-I", CSRC, "-o", out,
                                 os.path.join(CSRC, "mt_capi.hip")]
    subprocess.check_call(cmd)
    return out


def build_knobs(verbose: bool = False) -> str:
    """-DMT_DEBUG_KNOBS: the dump facilities (MT_DEBUG_ITEM_CYCLES, _PRINT_UNITS, _TIMELINE, _HEARTBEAT), read from
    the environment once in mt_scene_create.  A library of its own, lib/libmythtracer_hip_knobs.so, loaded by the
    experiment scripts that need the dumps (HipAbi(path)); never what tests or bench.py measure."""
    out = os.path.join(LIB, "libmythtracer_hip_knobs.so")
    cmd = [HIPCC] + HIP_FLAGS + ["-DMT_DEBUG_KNOBS"] + os.environ.get("MT_EXTRA_FLAGS", "").split() + [
        "-I", INC, "-I", CSRC, "-o", out, os.path.join(CSRC, "mt_capi.hip")]
    subprocess.check_call(cmd)
    return out


def build_host(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIB, exist_ok=True)
    srcs = _files(os.path.join(HOST, "src"), (".cc",))
    deps = srcs + _files(os.path.join(HOST, "include"), (".h",)) + [
        os.path.join(INC, "mythtracer_hip.h"), HIP_LIB]
    if not force and _newer(HOST_LIB, deps):
        return HOST_LIB
    cmd = (["g++"] + HOST_FLAGS + ["-I", INC, "-I", os.path.join(HOST, "include"),
                                   "-o", HOST_LIB] + srcs +
           ["-L", LIB, "-lmythtracer_hip", "-Wl,-rpath,$ORIGIN"])
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return HOST_LIB


def build_all(force: bool = False, verbose: bool = False):
    return build_hip(force, verbose), build_host(force, verbose)


if __name__ == "__main__":
    if "--diag" in sys.argv:
        print("built:", build_diag())
        sys.exit(0)
    if "--knobs" in sys.argv:
        print("built:", build_knobs())
        sys.exit(0)
    if "--prof" in sys.argv:
        print("built:", build_prof())
        sys.exit(0)
    build_all(force="--force" in sys.argv, verbose=True)
    print("built:", HIP_LIB, HOST_LIB)
