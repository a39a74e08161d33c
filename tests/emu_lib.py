"""Loader for tests/emu/libsf_emu.so: the device core run on a CPU wave emulator.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from strikeforce_amd import abi
from oracle_lib import ROOT, dump_with

_LIBS = {}


def lib(asan=False):
    name = "libsf_emu_asan.so" if asan else "libsf_emu.so"
    if name not in _LIBS:
        d = os.path.join(ROOT, "tests", "emu")
        if not os.path.exists(os.path.join(d, name)) or os.path.isdir("/root/reference"):
            # (re)built in the build container only (make is a no-op when up to date); elsewhere the file that
            # __graft_entry__.build() made travels with the snapshot and nothing compiles under the tests
            subprocess.check_call(["make", "-s", "-C", d, name])
        L = C.CDLL(os.path.join(d, name))
        L.sfe_create.argtypes = [C.POINTER(abi.Config)]
        L.sfe_create.restype = C.c_void_p
        abi.bind(L, "sfe_")
        L.sfe_step_many.argtypes = [C.c_void_p, C.c_char_p, C.c_int32]
        L.sfe_last_error.restype = C.c_char_p
        _LIBS[name] = L
    return _LIBS[name]


class Emu:
    """Same call surface as oracle_lib.Oracle."""

    def __init__(self, workload, asan=False):
        self.w = workload
        self.cfg = workload.cfg
        self.L = lib(asan)
        self.h = self.L.sfe_create(C.byref(self.cfg))
        if not self.h:
            raise ValueError("emu rejected the configuration: %s" % self.L.sfe_last_error().decode())

    def close(self):
        if self.h:
            self.L.sfe_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def reset(self, tb, serial):
        assert self.L.sfe_reset(self.h, tb, serial) == 0

    def step(self, cmd):
        cmd = np.ascontiguousarray(cmd, dtype=np.uint8)
        assert cmd.size == self.cfg.arenas * self.cfg.n_agents
        assert self.L.sfe_step(self.h, cmd.ctypes.data_as(C.c_char_p)) == 0

    def step_many(self, cmds):
        cmds = np.ascontiguousarray(cmds, dtype=np.uint8)
        assert self.L.sfe_step_many(self.h, cmds.ctypes.data_as(C.c_char_p), cmds.shape[0]) == 0

    def step_begin(self):
        assert self.L.sfe_step_begin(self.h) == 0

    def step_end(self, cmd):
        cmd = np.ascontiguousarray(cmd, dtype=np.uint8)
        assert cmd.size == self.cfg.arenas * self.cfg.n_agents
        assert self.L.sfe_step_end(self.h, cmd.ctypes.data_as(C.c_char_p)) == 0

    def agent_alive(self):
        out = np.zeros((self.cfg.arenas, self.cfg.n_agents), dtype=np.uint8)
        assert self.L.sfe_agent_alive(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
        return out

    def phase_draws(self, arena=None):
        out = np.zeros((self.cfg.arenas, 6), dtype=np.int32)
        assert self.L.sfe_phase_draws(self.h, out.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        return out if arena is None else [int(x) for x in out[arena]]

    def observe(self):
        out = np.empty((self.cfg.arenas, self.cfg.n_agents, abi.OBS_CHANNELS, abi.OBS_WINDOW, abi.OBS_WINDOW),
                       dtype=np.float32)
        assert self.L.sfe_observe(self.h, out.ctypes.data_as(C.POINTER(C.c_float))) == 0
        return out

    def results(self):
        out = np.zeros((self.cfg.arenas, self.cfg.n_agents, 8), dtype=np.int32)
        assert self.L.sfe_results(self.h, out.ctypes.data_as(C.POINTER(C.c_int32))) == 0
        return out

    def done(self):
        out = np.zeros(self.cfg.arenas, dtype=np.uint8)
        assert self.L.sfe_done(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8))) == 0
        return out

    def digest(self):
        out = np.zeros(self.cfg.arenas, dtype=np.uint64)
        assert self.L.sfe_state_digest(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
        return out

    def dump(self, arena):
        return dump_with(self.L.sfe_dump_arena, self.h, self.cfg, arena)
