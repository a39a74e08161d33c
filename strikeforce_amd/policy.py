"""ctypes plumbing for the on-device policy network (include/strikeforce_policy.h).

``PolicyBatch`` evaluates the reference's bot network (bots/bot-0.5/Modules.hpp:54-179) for every agent of an
arena batch on the GPU, straight from the observation buffer ``ArenaBatch.observe_device`` wrote.  Parameters are
passed as a dict of float32 numpy arrays keyed by the reference's parameter names (``named_parameters()`` of
``AgentModel``); ``init_parameters`` makes a random set with torch's default initialisers for the same shapes.
There is no CPU path: without the HIP library / a GPU this raises.
"""
import ctypes as C

import numpy as np

from . import env

HIDDEN, ACTIONS, CONVS, RES_LAYERS = 160, 9, 4, 3
OBS_CHANNELS, OBS_WINDOW = 32, 31
ACTION_STRING = "+xzqeawsd"  # gameplay::prepare, bots/bot-0.5/Custom.hpp:162
POLICY_ABI_VERSION = 1

_FP = C.POINTER(C.c_float)


class Weights(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("conv_w", _FP * CONVS),
        ("gru_w_ih", _FP * 2), ("gru_w_hh", _FP * 2), ("gru_b_ih", _FP * 2), ("gru_b_hh", _FP * 2),
        ("comb_w", _FP), ("comb_b", _FP),
        ("value_res_w", _FP * RES_LAYERS), ("value_res_b", _FP * RES_LAYERS), ("value_w", _FP), ("value_b", _FP),
        ("policy_res_w", _FP * RES_LAYERS), ("policy_res_b", _FP * RES_LAYERS), ("policy_w", _FP), ("policy_b", _FP),
    ]


def parameter_shapes():
    """name -> shape, the names and shapes of AgentModel's parameters (Modules.hpp:37,62,87-91,147-152)."""
    s = {}
    for i in range(CONVS):
        s["backbone.cnn.conv%d.weight" % i] = (HIDDEN, OBS_CHANNELS if i == 0 else HIDDEN, 3, 3)
    for g in range(2):
        s["backbone.gru%d.weight_ih_l0" % g] = (3 * HIDDEN, HIDDEN)
        s["backbone.gru%d.weight_hh_l0" % g] = (3 * HIDDEN, HIDDEN)
        s["backbone.gru%d.bias_ih_l0" % g] = (3 * HIDDEN,)
        s["backbone.gru%d.bias_hh_l0" % g] = (3 * HIDDEN,)
    s["backbone.combined_processor.0.weight"] = (HIDDEN, 2 * HIDDEN + ACTIONS)
    s["backbone.combined_processor.0.bias"] = (HIDDEN,)
    for head, n_out in (("value", 1), ("policy", ACTIONS)):
        for i in range(RES_LAYERS):
            s["%s.0.lin%d.weight" % (head, i)] = (HIDDEN, HIDDEN)
            s["%s.0.lin%d.bias" % (head, i)] = (HIDDEN,)
        s["%s.1.weight" % head] = (n_out, HIDDEN)
        s["%s.1.bias" % head] = (n_out,)
    return s


def init_parameters(seed=0, gain=1.0):
    """Random parameters of the reference's shapes: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like torch's defaults for
    Conv2d / Linear, U(-1/sqrt(hidden), 1/sqrt(hidden)) for GRU.  (There is no network to fetch checkpoints.)"""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in parameter_shapes().items():
        if ".gru" in name:
            bound = 1.0 / np.sqrt(HIDDEN)
        elif name.endswith(".bias"):
            w = parameter_shapes()[name[:-5] + ".weight"]
            bound = 1.0 / np.sqrt(np.prod(w[1:]))
        else:
            bound = 1.0 / np.sqrt(np.prod(shape[1:]))
        out[name] = (rng.uniform(-bound, bound, size=shape) * gain).astype(np.float32)
    return out


def load_checkpoint(path):
    """Parameters of a trained bot from the reference's checkpoint file: `torch::save(model, dir + "/model.pt")`
    (bots/bot-0.5/Agent.hpp:77-110,124,159-161) writes a TorchScript module archive whose named_parameters() carry the
    names AgentModel registered (Modules.hpp:37,62,87-91,147-152) — the names `parameter_shapes()` lists.  Also reads a
    plain state_dict saved from Python.  Returns {name: float32 numpy array}, every name and shape checked; nothing
    is filled in silently."""
    import torch
    named = None
    try:
        m = torch.jit.load(path, map_location="cpu")
        named = {k: v for k, v in m.named_parameters()}
    except Exception:  # noqa: BLE001  (not a TorchScript archive)
        obj = torch.load(path, map_location="cpu", weights_only=True)
        if hasattr(obj, "state_dict"):
            obj = obj.state_dict()
        if not isinstance(obj, dict):
            raise ValueError("%s holds neither a TorchScript module nor a state_dict" % path)
        named = obj
    out = {}
    for name, shape in parameter_shapes().items():
        if name not in named:
            raise ValueError("checkpoint %s lacks parameter %s" % (path, name))
        a = named[name].detach().to(torch.float32).contiguous().numpy()
        if a.shape != tuple(shape):
            raise ValueError("checkpoint parameter %s has shape %s, expected %s" % (name, a.shape, shape))
        out[name] = a.copy()
    extra = sorted(k for k in named if k not in out and not k.endswith("num_batches_tracked"))
    if extra:
        raise ValueError("checkpoint %s holds parameters this network does not have: %s" % (path, extra[:5]))
    return out


class PredictIO(C.Structure):
    """sf_policy_predict_io (include/strikeforce_policy.h)."""
    _fields_ = [("d_keys", C.c_void_p), ("d_vals", C.c_void_p), ("d_counts", C.c_void_p), ("d_pov", C.c_void_p), ("cap", C.c_int32),
                ("d_dense", C.c_void_p), ("d_reset_mask", C.c_void_p), ("d_reset_words", C.c_void_p),
                ("reset_stride", C.c_int32), ("reset_group", C.c_int32), ("action_string", C.c_char_p), ("seed", C.c_uint64),
                ("greedy", C.c_int32), ("d_probs", C.c_void_p), ("d_value", C.c_void_p), ("d_cmd", C.c_void_p), ("d_action", C.c_void_p)]


def _bind(L):
    if getattr(L, "_sf_policy_bound", False):
        return
    vp = C.c_void_p
    L.sf_policy_predict_sparse.argtypes = [vp, C.POINTER(PredictIO), C.c_int32]
    L.sf_policy_create.argtypes = [C.POINTER(Weights), C.c_int32, C.c_int32, C.POINTER(vp)]
    L.sf_policy_destroy.argtypes = [vp]
    L.sf_policy_destroy.restype = None
    L.sf_policy_reset_memory.argtypes = [vp, vp]
    L.sf_policy_reset_memory_n.argtypes = [vp, vp, C.c_int32]
    L.sf_policy_reset_memory_n.restype = C.c_int
    L.sf_policy_forward.argtypes = [vp, vp, C.c_int32, vp, vp]
    L.sf_policy_forward_sparse.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, vp, vp]
    L.sf_policy_forward_sparse_or_dense.argtypes = [vp, vp, vp, vp, vp, C.c_int32, C.c_int32, vp, vp, vp]
    L.sf_policy_kernel_time_by_kernel.argtypes = [vp, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.sf_policy_sparse_overflows.argtypes = [vp, C.POINTER(C.c_int32)]
    L.sf_policy_act.argtypes = [vp, vp, C.c_int32, C.c_char_p, C.c_uint64, C.c_int32, vp, vp]
    L.sf_policy_get_memory.argtypes = [vp, C.c_int32, _FP, _FP]
    L.sf_policy_set_memory.argtypes = [vp, C.c_int32, _FP, _FP]
    L.sf_policy_set_stream.argtypes = [vp, vp]
    L.sf_policy_synchronize.argtypes = [vp]
    L.sf_policy_gemm.argtypes = [vp, vp, C.c_int32, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.sf_policy_gemm_split.argtypes = L.sf_policy_gemm.argtypes
    L.sf_policy_features.argtypes = [vp, vp, C.c_int32, vp]
    L.sf_policy_kernel_time.argtypes = [vp, C.c_int32, _FP, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.sf_policy_kernel_time_ex.argtypes = L.sf_policy_kernel_time.argtypes
    for n in EXPORTS:
        if n != "sf_policy_destroy":
            getattr(L, n).restype = C.c_int
    L._sf_policy_bound = True


# every symbol include/strikeforce_policy.h declares
EXPORTS = ["sf_policy_create", "sf_policy_destroy", "sf_policy_reset_memory", "sf_policy_reset_memory_n", "sf_policy_forward", "sf_policy_forward_sparse",
           "sf_policy_forward_sparse_or_dense",
           "sf_policy_sparse_overflows", "sf_policy_act", "sf_policy_predict_sparse",
           "sf_policy_get_memory", "sf_policy_set_memory", "sf_policy_set_stream", "sf_policy_synchronize",
           "sf_policy_kernel_time", "sf_policy_kernel_time_ex", "sf_policy_kernel_time_by_kernel", "sf_policy_gemm", "sf_policy_gemm_split", "sf_policy_features", "sf_policy_abi_version"]


class PolicyBatch:
    """`max_agents` independent copies of the reference's AgentModel state over one shared parameter set."""

    def __init__(self, params, max_agents, device=0):
        self.L = env.load_library()
        _bind(self.L)
        self.max_agents = int(max_agents)
        shapes = parameter_shapes()
        keep = {}
        for name, shape in shapes.items():
            if name not in params:
                raise ValueError("missing parameter %s" % name)
            a = np.ascontiguousarray(params[name], dtype=np.float32)
            if a.shape != tuple(shape):
                raise ValueError("parameter %s has shape %s, expected %s" % (name, a.shape, shape))
            keep[name] = a
        ptr = lambda n: keep[n].ctypes.data_as(_FP)
        w = Weights()
        w.abi_version = POLICY_ABI_VERSION
        for i in range(CONVS):
            w.conv_w[i] = ptr("backbone.cnn.conv%d.weight" % i)
        for g in range(2):
            w.gru_w_ih[g] = ptr("backbone.gru%d.weight_ih_l0" % g)
            w.gru_w_hh[g] = ptr("backbone.gru%d.weight_hh_l0" % g)
            w.gru_b_ih[g] = ptr("backbone.gru%d.bias_ih_l0" % g)
            w.gru_b_hh[g] = ptr("backbone.gru%d.bias_hh_l0" % g)
        w.comb_w, w.comb_b = ptr("backbone.combined_processor.0.weight"), ptr("backbone.combined_processor.0.bias")
        for i in range(RES_LAYERS):
            w.value_res_w[i], w.value_res_b[i] = ptr("value.0.lin%d.weight" % i), ptr("value.0.lin%d.bias" % i)
            w.policy_res_w[i], w.policy_res_b[i] = ptr("policy.0.lin%d.weight" % i), ptr("policy.0.lin%d.bias" % i)
        w.value_w, w.value_b = ptr("value.1.weight"), ptr("value.1.bias")
        w.policy_w, w.policy_b = ptr("policy.1.weight"), ptr("policy.1.bias")
        self.h = C.c_void_p()
        rc = self.L.sf_policy_create(C.byref(w), self.max_agents, int(device), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise env.StrikeForceError("sf_policy_create failed (%d): %s" % (rc, self.L.sf_last_error().decode()))

    def _ck(self, rc, what):
        if rc != 0:
            raise env.StrikeForceError("%s failed (%d): %s" % (what, rc, self.L.sf_last_error().decode()))

    def close(self):
        if getattr(self, "h", None):
            self.L.sf_policy_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        self._ck(self.L.sf_policy_set_stream(self.h, C.c_void_p(hip_stream)), "sf_policy_set_stream")

    def synchronize(self):
        self._ck(self.L.sf_policy_synchronize(self.h), "sf_policy_synchronize")

    def reset_memory(self, d_mask_ptr=None, agents=None):
        """d_mask: device bytes, one per agent; with `agents` only agents [0, agents) are looked at (a mask of that
        many bytes), otherwise the mask must hold max_agents bytes."""
        m = C.c_void_p(d_mask_ptr) if d_mask_ptr else None
        if agents is None:
            self._ck(self.L.sf_policy_reset_memory(self.h, m), "sf_policy_reset_memory")
        else:
            self._ck(self.L.sf_policy_reset_memory_n(self.h, m, int(agents)), "sf_policy_reset_memory_n")

    def forward(self, d_obs_ptr, agents, d_probs_ptr, d_value_ptr):
        self._ck(self.L.sf_policy_forward(self.h, C.c_void_p(d_obs_ptr), int(agents), C.c_void_p(d_probs_ptr),
                                          C.c_void_p(d_value_ptr)), "sf_policy_forward")

    def forward_sparse(self, d_keys_ptr, d_vals_ptr, d_counts_ptr, d_pov_ptr, cap, agents, d_probs_ptr, d_value_ptr,
                       d_dense_ptr=None):
        """forward() on ArenaBatch.observe_sparse_device's lists: same results, no dense observation in between.
        d_dense_ptr: the buffer ArenaBatch.observe_overflow_device filled for the agents whose list did not fit; they are
        then evaluated from it (never on a blank window).  Without it such agents are evaluated as empty and counted
        (sparse_overflows)."""
        if d_dense_ptr is not None:
            self._ck(self.L.sf_policy_forward_sparse_or_dense(
                self.h, C.c_void_p(d_keys_ptr), C.c_void_p(d_vals_ptr), C.c_void_p(d_counts_ptr), C.c_void_p(d_pov_ptr),
                int(cap), int(agents), C.c_void_p(d_dense_ptr), C.c_void_p(d_probs_ptr), C.c_void_p(d_value_ptr)),
                "sf_policy_forward_sparse_or_dense")
            return
        self._ck(self.L.sf_policy_forward_sparse(self.h, C.c_void_p(d_keys_ptr), C.c_void_p(d_vals_ptr), C.c_void_p(d_counts_ptr),
                                                 C.c_void_p(d_pov_ptr), int(cap), int(agents), C.c_void_p(d_probs_ptr),
                                                 C.c_void_p(d_value_ptr)), "sf_policy_forward_sparse")

    def sparse_overflows(self):
        """Agents evaluated on an empty observation since the last call because their list did not fit (synchronises)."""
        n = C.c_int32()
        self._ck(self.L.sf_policy_sparse_overflows(self.h, C.byref(n)), "sf_policy_sparse_overflows")
        return n.value

    def predict_sparse(self, d_keys_ptr, d_vals_ptr, d_counts_ptr, d_pov_ptr, cap, agents, d_probs_ptr, d_value_ptr, d_cmd_ptr,
                       seed=0, greedy=False, d_action_ptr=None, action_string=ACTION_STRING, d_dense_ptr=None, d_reset_mask_ptr=None,
                       reset_words=None):
        """reset_memory(mask) + forward_sparse + act as the forward's two launches (Agent::predict + update are one call in
        the reference too): same results bit for bit.  reset_words: ArenaBatch.done_view_device()'s triple, read in place."""
        io = PredictIO()
        io.d_keys, io.d_vals, io.d_counts, io.d_pov, io.cap = d_keys_ptr, d_vals_ptr, d_counts_ptr, d_pov_ptr, int(cap)
        io.d_dense, io.d_reset_mask = d_dense_ptr, d_reset_mask_ptr
        if reset_words is not None:
            io.d_reset_words, io.reset_stride, io.reset_group = reset_words
        io.action_string, io.seed, io.greedy = action_string.encode(), seed, 1 if greedy else 0
        io.d_probs, io.d_value, io.d_cmd, io.d_action = d_probs_ptr, d_value_ptr, d_cmd_ptr, d_action_ptr
        self._ck(self.L.sf_policy_predict_sparse(self.h, C.byref(io), int(agents)), "sf_policy_predict_sparse")

    def act(self, d_probs_ptr, agents, d_cmd_ptr, seed=0, greedy=False, d_action_ptr=None,
            action_string=ACTION_STRING):
        self._ck(self.L.sf_policy_act(self.h, C.c_void_p(d_probs_ptr), int(agents), action_string.encode(),
                                      C.c_uint64(seed), 1 if greedy else 0, C.c_void_p(d_cmd_ptr),
                                      C.c_void_p(d_action_ptr) if d_action_ptr else None), "sf_policy_act")

    def get_memory(self, agent):
        h = np.zeros((2, HIDDEN), dtype=np.float32)
        a = np.zeros(ACTIONS, dtype=np.float32)
        self._ck(self.L.sf_policy_get_memory(self.h, int(agent), h.ctypes.data_as(_FP), a.ctypes.data_as(_FP)),
                 "sf_policy_get_memory")
        return h, a

    def set_memory(self, agent, h, action_input):
        h = np.ascontiguousarray(h, dtype=np.float32).reshape(2, HIDDEN)
        a = np.ascontiguousarray(action_input, dtype=np.float32).reshape(ACTIONS)
        self._ck(self.L.sf_policy_set_memory(self.h, int(agent), h.ctypes.data_as(_FP), a.ctypes.data_as(_FP)),
                 "sf_policy_set_memory")

    def gemm(self, d_a_ptr, lda, d_w_ptr, d_bias_ptr, d_c_ptr, ldc, m, n, k):
        """The MFMA matrix kernel on its own: C = A W^T + bias on device buffers."""
        self._ck(self.L.sf_policy_gemm(self.h, C.c_void_p(d_a_ptr), lda, C.c_void_p(d_w_ptr),
                                       C.c_void_p(d_bias_ptr) if d_bias_ptr else None, C.c_void_p(d_c_ptr), ldc, m, n, k),
                 "sf_policy_gemm")

    def features(self, d_obs_ptr, agents, d_feat_ptr):
        """GameCNN::forward alone: [agents][160] features behind the four convolutions (test hook)."""
        self._ck(self.L.sf_policy_features(self.h, C.c_void_p(d_obs_ptr), agents, C.c_void_p(d_feat_ptr)), "sf_policy_features")

    def gemm_split(self, d_a_ptr, lda, d_w_ptr, d_bias_ptr, d_c_ptr, ldc, m, n, k):
        """The same product through the bf16-split kernel of conv1 / conv2 (six bf16 MFMAs per block, f32-level error)."""
        self._ck(self.L.sf_policy_gemm_split(self.h, C.c_void_p(d_a_ptr), lda, C.c_void_p(d_w_ptr),
                                             C.c_void_p(d_bias_ptr) if d_bias_ptr else None, C.c_void_p(d_c_ptr), ldc, m, n, k),
                 "sf_policy_gemm_split")

    def kernel_time_by_pipe(self, enable=True):
        """[(ms, flop, launches) of the f32-MFMA launches, (...) of the bf16-split launches] since the last call."""
        ms, fl, n = (C.c_float * 2)(), (C.c_double * 2)(), (C.c_int32 * 2)()
        self._ck(self.L.sf_policy_kernel_time_ex(self.h, 1 if enable else 0, ms, fl, n), "sf_policy_kernel_time_ex")
        return [(ms[k], fl[k], n[k]) for k in range(2)]

    def kernel_time_by_kernel(self, enable=True):
        """[(ms, algorithmic flop, launches)] for k_gemm (f32 MFMA), k_gemm_b3 (bf16 split), conv0 on the non-zeros, k_tail."""
        ms, fl, n = (C.c_float * 4)(), (C.c_double * 4)(), (C.c_int32 * 4)()
        self._ck(self.L.sf_policy_kernel_time_by_kernel(self.h, 1 if enable else 0, ms, fl, n), "sf_policy_kernel_time_by_kernel")
        return [(float(ms[k]), float(fl[k]), int(n[k])) for k in range(4)]

    def kernel_time(self, enable=True):
        """(ms, flop, launches) of the MFMA GEMM launches since the last call; arms / disarms the timers."""
        ms, fl, n = C.c_float(), C.c_double(), C.c_int32()
        self._ck(self.L.sf_policy_kernel_time(self.h, 1 if enable else 0, C.byref(ms), C.byref(fl), C.byref(n)),
                 "sf_policy_kernel_time")
        return ms.value, fl.value, n.value
