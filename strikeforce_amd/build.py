"""Builds libstrikeforce_amd.so (gfx950 kernels + C-ABI) in-tree with hipcc.

    python -m strikeforce_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels with the snapshot
to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstrikeforce_amd.so")
SOURCES = ["sf_api.hip", "sf_policy.hip"]
DEPS = ["sf_api.hip", "sf_policy.hip", "sf_core.hpp", "sf_obs.hpp", "sf_host.hpp", "sf_types.hpp", "wave_gfx950.hpp",
        os.path.join("..", "..", "include", "strikeforce.h"),
        os.path.join("..", "..", "include", "strikeforce_policy.h")]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=True):
    if not force and not stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB] + SOURCES
    if verbose:
        print("[strikeforce_amd.build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
