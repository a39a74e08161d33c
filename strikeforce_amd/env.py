"""Host-side mirror of the reference's environment surface over the C-ABI (include/strikeforce.h).

``ArenaBatch`` is plumbing only: it loads libstrikeforce_amd.so (hand-written gfx950 kernels) and moves
pointers.  There is no CPU path — if the library is missing or no MI355X is visible it raises.
"""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# SF_LIBRARY_PATH: load another build of the same library (A/B timing of two builds inside one gpurun call)
LIB_PATH = os.environ.get("SF_LIBRARY_PATH") or os.path.join(_HERE, "libstrikeforce_amd.so")
_LIB = None


class StrikeForceError(RuntimeError):
    pass


def load_library():
    """Load the HIP library and declare its signatures.  Raises if it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise StrikeForceError(
                "libstrikeforce_amd.so is missing: run `python -m strikeforce_amd.build` (hipcc, gfx950). "
                "strikeforce_amd has no CPU path.")
        # PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64.  If this library's copies (from /opt/rocm)
        # were mapped first, a later `import torch` would bring a second HSA runtime into the process and
        # torch.cuda would find no GPU.  Importing torch first makes the loader resolve our DT_NEEDED entries to
        # the runtime that is already mapped (same sonames), so there is exactly one runtime either way.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.sf_create.argtypes = [C.POINTER(abi.Config), C.POINTER(vp)]
        L.sf_create.restype = C.c_int
        abi.bind(L, "sf_")
        L.sf_destroy.restype = C.c_int
        L.sf_step_device.argtypes = [vp, vp, C.c_int32]
        L.sf_observe_device.argtypes = [vp, vp]
        L.sf_observe_device_delta.argtypes = [vp, vp]
        L.sf_observe_sparse_device.argtypes = [vp, vp, vp, vp, vp, C.c_int32]
        L.sf_observe_overflow_device.argtypes = [vp, vp, C.c_int32, vp, vp]
        L.sf_observe_overflow_device.restype = C.c_int
        L.sf_results_device.argtypes = [vp, vp]
        L.sf_done_device.argtypes = [vp, vp]
        if hasattr(L, "sf_done_view_device"):  # (A/B runs against older builds of the library, tools/ab.sh)
            L.sf_done_view_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
            L.sf_done_view_device.restype = C.c_int
        L.sf_set_stream.argtypes = [vp, vp]
        L.sf_synchronize.argtypes = [vp]
        L.sf_kernel_time.argtypes = [vp, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.sf_comm_unique_id.argtypes = [C.c_char_p]
        L.sf_comm_init.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32]
        L.sf_results_allgather.argtypes = [vp, vp]
        L.sf_comm_wait.argtypes = [vp, C.c_int32]
        L.sf_comm_ranks.argtypes = [vp, C.POINTER(C.c_int32)]
        L.sf_step_end_device.argtypes = [vp, vp]
        L.sf_agent_alive_device.argtypes = [vp, vp]
        for n in ("sf_comm_unique_id", "sf_comm_init", "sf_results_allgather", "sf_comm_wait", "sf_comm_ranks",
                  "sf_step_end_device", "sf_agent_alive_device"):
            getattr(L, n).restype = C.c_int
        L.sf_config_defaults.argtypes = [C.POINTER(abi.Config)]
        L.sf_config_defaults.restype = None
        for n in ("sf_step_device", "sf_observe_device", "sf_observe_device_delta", "sf_observe_sparse_device", "sf_results_device", "sf_done_device", "sf_set_stream", "sf_synchronize",
                  "sf_kernel_time", "sf_abi_version"):
            getattr(L, n).restype = C.c_int
        L.sf_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


# every symbol include/strikeforce.h declares
EXPORTS = ["sf_create", "sf_destroy", "sf_config_defaults", "sf_reset", "sf_step", "sf_step_device", "sf_observe",
           "sf_observe_device", "sf_observe_device_delta", "sf_observe_sparse_device", "sf_observe_overflow_device", "sf_results", "sf_results_device", "sf_done", "sf_done_device", "sf_done_view_device", "sf_state_digest", "sf_dump_arena",
           "sf_set_stream", "sf_synchronize", "sf_kernel_time", "sf_last_error", "sf_abi_version",
           "sf_comm_unique_id", "sf_comm_init", "sf_results_allgather", "sf_comm_wait", "sf_comm_ranks",
           "sf_step_begin", "sf_step_end", "sf_step_end_device", "sf_agent_alive", "sf_agent_alive_device", "sf_phase_draws"]


class ArenaBatch:
    """A batch of independent arenas on one GPU.

    Mirrors the reference loop (gameplay.hpp:1428-1505): ``reset`` = setup()+load_data() and the first
    loop top; ``step`` = one iteration of play()'s while(true) for every arena, commands are the
    reference's command chars; ``observe`` = gameplay::bot()'s 32x31x31 encoding
    (bots/bot-0.5/Custom.hpp:137-159); ``done`` = check_end().
    """

    def __init__(self, workload):
        self.w = workload
        self.cfg = workload.cfg
        self.L = load_library()
        self.h = C.c_void_p()
        rc = self.L.sf_create(C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise StrikeForceError("sf_create failed (%d): %s" % (rc, self.L.sf_last_error().decode()))

    def _ck(self, rc, what):
        if rc != 0:
            raise StrikeForceError("%s failed (%d): %s" % (what, rc, self.L.sf_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.sf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        self._ck(self.L.sf_set_stream(self.h, C.c_void_p(hip_stream)), "sf_set_stream")

    def synchronize(self):
        self._ck(self.L.sf_synchronize(self.h), "sf_synchronize")

    def reset(self, tb, serial):
        self._ck(self.L.sf_reset(self.h, tb, serial), "sf_reset")

    def step(self, cmd):
        cmd = np.ascontiguousarray(cmd, dtype=np.uint8)
        if cmd.size != self.cfg.arenas * self.cfg.n_agents:
            raise ValueError("cmd must hold arenas * n_agents command chars")
        self._ck(self.L.sf_step(self.h, cmd.ctypes.data_as(C.c_char_p)), "sf_step")

    def step_begin(self):
        """The first half of one iteration (gameplay.hpp:1455-1463): up to where the reference asks the agents of humans
        other than `ind` for their command."""
        self._ck(self.L.sf_step_begin(self.h), "sf_step_begin")

    def step_end(self, cmd):
        """The second half (gameplay.hpp:1464-1471) and the next loop top; cmd as for step()."""
        cmd = np.ascontiguousarray(cmd, dtype=np.uint8)
        if cmd.size != self.cfg.arenas * self.cfg.n_agents:
            raise ValueError("cmd must hold arenas * n_agents command chars")
        self._ck(self.L.sf_step_end(self.h, cmd.ctypes.data_as(C.c_char_p)), "sf_step_end")

    def step_end_device(self, d_cmd_ptr):
        self._ck(self.L.sf_step_end_device(self.h, C.c_void_p(d_cmd_ptr)), "sf_step_end_device")

    def agent_alive(self):
        """[arenas][n_agents] uint8: 1 while the commanded human is alive and still has its Agent (Human::active_agent)."""
        out = np.zeros((self.cfg.arenas, self.cfg.n_agents), dtype=np.uint8)
        self._ck(self.L.sf_agent_alive(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8))), "sf_agent_alive")
        return out

    def phase_draws(self, arena=None):
        """Generator draws of the last step by phase [zombie_action, update_bull, human_action, update_bull, spawns, rest]:
        of one arena (a list of six ints) or of all ([arenas][6] int32)."""
        out = np.zeros((self.cfg.arenas, 6), dtype=np.int32)
        self._ck(self.L.sf_phase_draws(self.h, out.ctypes.data_as(C.POINTER(C.c_int32))), "sf_phase_draws")
        return out if arena is None else [int(x) for x in out[arena]]

    def agent_alive_device(self, d_out_ptr):
        self._ck(self.L.sf_agent_alive_device(self.h, C.c_void_p(d_out_ptr)), "sf_agent_alive_device")

    def comm_ranks(self):
        n = C.c_int32(0)
        self._ck(self.L.sf_comm_ranks(self.h, C.byref(n)), "sf_comm_ranks")
        return int(n.value)

    def step_device(self, d_cmd_ptr, k):
        """k loop iterations in one launch; d_cmd_ptr: device address of uint8 [k][arenas][n_agents]."""
        self._ck(self.L.sf_step_device(self.h, C.c_void_p(d_cmd_ptr), k), "sf_step_device")

    def observe(self):
        out = np.empty((self.cfg.arenas, self.cfg.n_agents, abi.OBS_CHANNELS, abi.OBS_WINDOW, abi.OBS_WINDOW),
                       dtype=np.float32)
        self._ck(self.L.sf_observe(self.h, out.ctypes.data_as(C.POINTER(C.c_float))), "sf_observe")
        return out

    def observe_device_delta(self, d_out_ptr):
        """observe_device for ONE persistent, caller-untouched buffer per env: only what changed is written."""
        self._ck(self.L.sf_observe_device_delta(self.h, C.c_void_p(d_out_ptr)), "sf_observe_device_delta")

    def observe_sparse_device(self, d_keys_ptr, d_vals_ptr, d_counts_ptr, d_pov_ptr, cap):
        """The observation as the list of its non-zero floats ([agents][cap] keys and values, [agents] counts,
        [agents][160] centre values): what PolicyBatch.forward_sparse consumes."""
        self._ck(self.L.sf_observe_sparse_device(self.h, C.c_void_p(d_keys_ptr), C.c_void_p(d_vals_ptr), C.c_void_p(d_counts_ptr),
                                                 C.c_void_p(d_pov_ptr), int(cap)), "sf_observe_sparse_device")

    def observe_overflow_device(self, d_counts_ptr, cap, d_dense_ptr, d_pov_ptr):
        """Right behind observe_sparse_device, same counts / cap: the dense observation of exactly the agents whose list
        did not fit, into their rows of d_dense ([agents][32][31][31]); their d_pov rows are rewritten from it."""
        self._ck(self.L.sf_observe_overflow_device(self.h, C.c_void_p(d_counts_ptr), int(cap), C.c_void_p(d_dense_ptr),
                                                   C.c_void_p(d_pov_ptr)), "sf_observe_overflow_device")

    def observe_device(self, d_out_ptr):
        self._ck(self.L.sf_observe_device(self.h, C.c_void_p(d_out_ptr)), "sf_observe_device")

    def results(self):
        out = np.zeros((self.cfg.arenas, self.cfg.n_agents, 8), dtype=np.int32)
        self._ck(self.L.sf_results(self.h, out.ctypes.data_as(C.POINTER(C.c_int32))), "sf_results")
        return out

    def results_device(self, d_out_ptr):
        self._ck(self.L.sf_results_device(self.h, C.c_void_p(d_out_ptr)), "sf_results_device")

    @staticmethod
    def comm_unique_id():
        """128 opaque bytes from rank 0 (ncclGetUniqueId through the library's own RCCL): hand them to every rank."""
        L = load_library()
        buf = C.create_string_buffer(128)
        rc = L.sf_comm_unique_id(buf)
        if rc != 0:
            raise StrikeForceError("sf_comm_unique_id failed (%d): %s" % (rc, L.sf_last_error().decode()))
        return buf.raw

    def comm_init(self, uid, rank, world):
        self._ck(self.L.sf_comm_init(self.h, uid, rank, world), "sf_comm_init")

    def results_allgather(self, d_out_ptr):
        """RCCL all-gather of every rank's result records into [world][arenas][agents][8] int32, on a side stream."""
        self._ck(self.L.sf_results_allgather(self.h, C.c_void_p(d_out_ptr)), "sf_results_allgather")

    def comm_wait(self, host_too=False):
        self._ck(self.L.sf_comm_wait(self.h, 1 if host_too else 0), "sf_comm_wait")

    def done_device(self, d_out_ptr):
        """check_end()'s verdict on the device, one byte per (arena, agent): what PolicyBatch.reset_memory takes."""
        self._ck(self.L.sf_done_device(self.h, C.c_void_p(d_out_ptr)), "sf_done_device")

    def done_view_device(self):
        """The same flags where the library keeps them: (device pointer, int32 stride, agents per arena) — what
        PolicyBatch.predict_sparse reads in place (reset_words=)."""
        ptr, stride, group = C.c_void_p(), C.c_int32(), C.c_int32()
        self._ck(self.L.sf_done_view_device(self.h, C.byref(ptr), C.byref(stride), C.byref(group)), "sf_done_view_device")
        return ptr.value, stride.value, group.value

    def done(self):
        out = np.zeros(self.cfg.arenas, dtype=np.uint8)
        self._ck(self.L.sf_done(self.h, out.ctypes.data_as(C.POINTER(C.c_uint8))), "sf_done")
        return out

    def digest(self):
        out = np.zeros(self.cfg.arenas, dtype=np.uint64)
        self._ck(self.L.sf_state_digest(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64))), "sf_state_digest")
        return out

    def dump_raw(self, arena):
        cfg = self.cfg
        cells = cfg.floors * cfg.rows * cfg.cols
        hdr = abi.ArenaHdr()
        hs = (abi.HumanRec * cfg.cap_humans)()
        zs = (abi.ZombieRec * cfg.cap_zombies)()
        bs = (abi.BulletRec * cfg.cap_bullets)()
        ps = (abi.PortalRec * cfg.cap_portals)()
        flags = np.zeros(cells, dtype=np.uint8)
        dmg = np.zeros(cells, dtype=np.int32)
        pidx = np.zeros(cells, dtype=np.int32)
        self._ck(self.L.sf_dump_arena(self.h, arena, C.byref(hdr), hs, zs, bs, ps,
                                      flags.ctypes.data_as(C.POINTER(C.c_uint8)),
                                      dmg.ctypes.data_as(C.POINTER(C.c_int32)),
                                      pidx.ctypes.data_as(C.POINTER(C.c_int32))), "sf_dump_arena")
        return hdr, list(hs), list(zs), list(bs), list(ps), flags, dmg, pidx

    def dump(self, arena):
        """dump_raw() as an object with named fields (hdr, humans, zombies, bullets, portals, flags, dmg, pidx)."""
        import types
        hdr, hs, zs, bs, ps, flags, dmg, pidx = self.dump_raw(arena)
        return types.SimpleNamespace(hdr=hdr, humans=hs, zombies=zs, bullets=bs, portals=ps, flags=flags, dmg=dmg,
                                     pidx=pidx)

    def kernel_time(self, enable=True):
        """(ms, launches) of the step kernels since the last call, from HIP events on the launch stream."""
        ms, n = C.c_float(0), C.c_int32(0)
        self._ck(self.L.sf_kernel_time(self.h, 1 if enable else 0, C.byref(ms), C.byref(n)), "sf_kernel_time")
        return ms.value, n.value
