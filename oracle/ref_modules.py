#!/usr/bin/env python3
"""ref_modules.py — TEST INFRASTRUCTURE: build oracle/_ref/libsf_refmodules.so, the reference's own bot network.

    python oracle/ref_modules.py [--ref /root/reference] [--quiet]

StrikeForce-client/bots/bot-0.5/Modules.hpp:26-180 (`#include <torch/torch.h>`, `#define LAYER_INDEX 3`, ResB, GameCNN,
Backbone, AgentModel) needs nothing but libtorch, and the torch wheel of this image ships libtorch's C++ headers and
libraries.  The range is cut out by line number into a temporary directory (anchor texts on its first and last line and
on the lines around it: a reference whose lines have moved fails loudly), compiled unedited together with
oracle/ref_modules_wrap.cpp (ours: C entry points only) against that libtorch, and the result is
oracle/_ref/libsf_refmodules.so (git-ignored).  Line 25 of the file, `#include "../../basic.hpp"`, is outside the range:
it reaches SFML (basic.hpp:41) and nothing in the range uses it.  The temporary directory is deleted; no reference source
stays in the repo.  Without a checkout, or without libtorch headers, nothing is built (exit code 0).

This pins SURVEY §8 row f-4's checker (oracle/policy_ref.py) on the reference itself: tests/test_ref_modules.py, and
tests/golden/make_policy_vectors.py writes tests/golden/policy_vectors.json from this library's outputs.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REL = "StrikeForce-client/bots/bot-0.5/Modules.hpp"
FIRST, LAST = 26, 180
ANCHORS = [
    (25, '#include "../../basic.hpp"'),     # the line right before the range (left out)
    (26, "#include <torch/torch.h>"),
    (28, "#define LAYER_INDEX 3"),
    (30, "struct ResBImpl : torch::nn::Module {"),
    (138, "struct AgentModelImpl : torch::nn::Module {"),
    (180, "TORCH_MODULE(AgentModel);"),
]
OUT = os.path.join(HERE, "_ref", "libsf_refmodules.so")


def torch_flags():
    """(include dirs, lib dir, abi flag) of the libtorch inside the torch wheel, or None."""
    try:
        import torch
    except Exception:  # noqa: BLE001
        return None
    tdir = os.path.dirname(torch.__file__)
    inc = [os.path.join(tdir, "include"), os.path.join(tdir, "include", "torch", "csrc", "api", "include")]
    if not os.path.exists(os.path.join(inc[1], "torch", "torch.h")):
        return None
    return inc, os.path.join(tdir, "lib"), int(torch.compiled_with_cxx11_abi())


def build(ref="/root/reference", quiet=False):
    src = os.path.join(ref, REL)
    if not os.path.isfile(src):
        if not quiet:
            print("no reference checkout at %s: oracle/_ref/libsf_refmodules.so left as is" % ref)
        return None
    fl = torch_flags()
    if fl is None:
        if not quiet:
            print("no libtorch C++ headers in this torch build: oracle/_ref/libsf_refmodules.so not built")
        return None
    wrap = os.path.join(HERE, "ref_modules_wrap.cpp")
    if os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(p) for p in (src, wrap, __file__)):
        return OUT
    with open(src, encoding="utf-8", errors="replace", newline="") as f:
        lines = f.read().split("\n")
    if len(lines) != LAST:
        raise SystemExit("ref_modules: %s has %d lines, expected %d — the reference has changed" % (REL, len(lines), LAST))
    for no, text in ANCHORS:
        if lines[no - 1].strip() != text:
            raise SystemExit("ref_modules: %s:%d is %r, expected %r — the reference's lines have moved"
                             % (REL, no, lines[no - 1].strip(), text))
    inc, libdir, abi = fl
    tmp = tempfile.mkdtemp(prefix="sf_refmodules_")
    try:
        with open(os.path.join(tmp, "slice_modules.inc"), "w", encoding="utf-8", newline="") as f:
            f.write("\n".join(lines[FIRST - 1:LAST]) + "\n")
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        cmd = ["g++", "-std=c++17", "-O1", "-w", "-fPIC", "-shared", "-D_GLIBCXX_USE_CXX11_ABI=%d" % abi,
               "-I", tmp, "-I" + inc[0], "-I" + inc[1], wrap, "-L" + libdir, "-ltorch", "-ltorch_cpu", "-lc10",
               "-Wl,-rpath," + libdir, "-o", OUT]
        subprocess.check_call(cmd)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not quiet:
        print("built oracle/_ref/libsf_refmodules.so from %s (bots/bot-0.5/Modules.hpp:%d-%d)" % (ref, FIRST, LAST))
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default=os.environ.get("REF", "/root/reference"))
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    build(a.ref, a.quiet)
    sys.exit(0)
