"""Loader for oracle/_ref/libsf_refmodules.so: the reference's own bot network (bots/bot-0.5/Modules.hpp:26-180 compiled
unedited against the libtorch inside the torch wheel, oracle/ref_modules.py).  Test infrastructure only.  lib() is None
where the file was never built (no checkout) or cannot be loaded (another torch build)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libsf_refmodules.so")
_LIB = []
F32P = C.POINTER(C.c_float)


def lib():
    if not _LIB:
        L = None
        if os.path.exists(PATH):
            try:
                import torch  # noqa: F401  (libtorch's libraries first: the .so resolves against the loaded ones)
                L = C.CDLL(PATH)
            except OSError:
                L = None
        if L is not None:
            L.rm_create.restype = C.c_void_p
            L.rm_destroy.argtypes = [C.c_void_p]
            L.rm_destroy.restype = None
            L.rm_param_count.argtypes = [C.c_void_p]
            L.rm_param_info.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_longlong)]
            L.rm_set_param.argtypes = [C.c_void_p, C.c_char_p, F32P, C.c_longlong]
            L.rm_forward.argtypes = [C.c_void_p, F32P, F32P, F32P, F32P]
            L.rm_update_actions.argtypes = [C.c_void_p, C.c_int]
            L.rm_update_actions.restype = None
            L.rm_reset_memory.argtypes = [C.c_void_p]
            L.rm_reset_memory.restype = None
        _LIB.append(L)
    return _LIB[0]


class RefAgentModel:
    """One reference AgentModel (its own h_state / action_input, like one reference Agent)."""

    def __init__(self, params=None):
        self.L = lib()
        self.h = C.c_void_p(self.L.rm_create())
        if not self.h:
            raise RuntimeError("rm_create failed")
        if params is not None:
            self.load(params)

    def close(self):
        if self.h:
            self.L.rm_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def parameters(self):
        """{name: shape} as the reference's named_parameters() reports them, in its order."""
        out = {}
        buf = C.create_string_buffer(128)
        shape = (C.c_longlong * 4)()
        for i in range(self.L.rm_param_count(self.h)):
            nd = self.L.rm_param_info(self.h, i, buf, 128, shape)
            out[buf.value.decode()] = tuple(int(shape[d]) for d in range(nd))
        return out

    def load(self, params):
        for k, v in params.items():
            a = np.ascontiguousarray(v, dtype=np.float32)
            rc = self.L.rm_set_param(self.h, k.encode(), a.ctypes.data_as(F32P), a.size)
            if rc != 0:
                raise ValueError("rm_set_param(%s) -> %d" % (k, rc))

    def forward(self, obs):
        """obs [32,31,31] f32 -> (probs[9], value, h[2,160]) of the reference's AgentModel::forward."""
        x = np.ascontiguousarray(obs, dtype=np.float32).reshape(-1)
        assert x.size == 32 * 31 * 31
        p, v, h = np.zeros(9, np.float32), np.zeros(1, np.float32), np.zeros((2, 160), np.float32)
        rc = self.L.rm_forward(self.h, x.ctypes.data_as(F32P), p.ctypes.data_as(F32P), v.ctypes.data_as(F32P),
                               h.ctypes.data_as(F32P))
        if rc != 0:
            raise RuntimeError("rm_forward -> %d" % rc)
        return p, float(v[0]), h

    def update_actions(self, action):
        self.L.rm_update_actions(self.h, int(action))

    def reset_memory(self):
        self.L.rm_reset_memory(self.h)
