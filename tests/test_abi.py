"""The C-ABI library: builds for gfx950, loads, exports every symbol include/strikeforce.h declares,
and refuses to run without a GPU (no CPU fallback in the product)."""
import ctypes as C
import os
import re

import pytest

from strikeforce_amd import abi, build, config, env

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_and_exports_header_symbols():
    build.build(verbose=False)
    L = env.load_library()
    header = open(os.path.join(ROOT, "include", "strikeforce.h")).read()
    declared = set(re.findall(r"\b(sf_[a-z_]+)\s*\(", header))
    declared -= {"sf_env"}
    assert declared == set(env.EXPORTS), declared ^ set(env.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert L.sf_abi_version() == abi.SF_ABI_VERSION


def test_struct_sizes_match_header():
    # layout pinned against include/strikeforce.h (sizes computed from the C declarations)
    assert C.sizeof(abi.Profile) == 32 * 4
    assert C.sizeof(abi.Items) == (12 + 16 + 32) * 4
    assert C.sizeof(abi.HumanRec) == 28 * 4
    assert C.sizeof(abi.ZombieRec) == 7 * 4
    assert C.sizeof(abi.BulletRec) == 11 * 4
    assert C.sizeof(abi.PortalRec) == 4 * 4
    assert C.sizeof(abi.ArenaHdr) == 10 * 8 + 20 * 4


def test_config_defaults_are_the_shipped_tables():
    L = env.load_library()
    cfg = abi.Config()
    L.sf_config_defaults(C.byref(cfg))
    ours = config.default_items()
    assert bytes(cfg.items) == bytes(ours)
    assert bytes(cfg.player) == bytes(abi.Profile.from_tokens(config.HUMAN_TOKENS))
    assert bytes(cfg.npc) == bytes(abi.Profile.from_tokens(config.HUMAN_ENEMY_TOKENS))


def test_rejects_bad_arguments_without_touching_a_device():
    L = env.load_library()
    h = C.c_void_p()
    w = config.baseline_workload("C1")
    w.cfg.level = 11
    assert L.sf_create(C.byref(w.cfg), C.byref(h)) == -1
    assert b"level" in L.sf_last_error()
    w = config.baseline_workload("C1")
    w.cfg.cap_bullets = 1000
    assert L.sf_create(C.byref(w.cfg), C.byref(h)) == -1


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    with pytest.raises(env.StrikeForceError, match="no HIP device"):
        env.ArenaBatch(config.baseline_workload("C1"))
