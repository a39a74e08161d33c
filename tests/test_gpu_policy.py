"""The on-device policy network (include/strikeforce_policy.h, f32 MFMA kernels) against the PyTorch f32 restatement
of the reference's model (oracle/policy_ref.py).  Floating point: the kernels differ from torch only in summation
order, so the gate is a relative tolerance, stated per test.

Two forms of the convolution stack run here (fixture `cnn`): "folded" — the default: the four bias-free convolutions of
GameCNN (Modules.hpp:66-71, nothing between them) composed into one matrix at sf_policy_create and applied to the
observation's non-zeros — and "layered" (SF_POLICY_LAYERED=1): the four convolutions one after the other as the
reference evaluates them.  Both are held to the same tolerance against the same reference outputs."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle_lib import Oracle
from strikeforce_amd import config, env, policy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import policy_ref  # noqa: E402

pytestmark = pytest.mark.gpu

# |hip - torch| <= ATOL + RTOL * |torch| on probabilities, value and the recurrent state
RTOL, ATOL = 5e-5, 1e-6


def _obs(rng, B):
    x = rng.uniform(0.0, 2.0, size=(B, 32, 31, 31)).astype(np.float32)
    x *= rng.uniform(size=x.shape) < 0.3
    x *= np.where(rng.uniform(size=x.shape) < 0.2, -1.0, 1.0).astype(np.float32)
    return x


def _dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    torch.cuda.synchronize()
    return t


def _memory(pb, B):
    h = np.zeros((2, B, 160), dtype=np.float32)
    a = np.zeros((B, 9), dtype=np.float32)
    for b in range(B):
        hb, ab = pb.get_memory(b)
        h[:, b], a[b] = hb, ab
    return h, a


@pytest.mark.parametrize("B,steps", [(1, 3), (5, 4), (37, 3), (100, 2), (300, 2), (400, 2)])
def test_forward_matches_reference(B, steps, cnn):
    """B picks the GEMM variant of each layer: 1-wave, 2-wave and 4-wave blocks, ragged last tiles."""
    rng = np.random.default_rng(B)
    params = policy.init_parameters(seed=B)
    pb = policy.PolicyBatch(params, B)
    h = np.zeros((2, B, 160), dtype=np.float32)
    a = np.zeros((B, 9), dtype=np.float32)
    a[:, 0] = 1
    d_probs = torch.zeros((B, 9), dtype=torch.float32, device="cuda")
    d_value = torch.zeros(B, dtype=torch.float32, device="cuda")
    d_cmd = torch.zeros(B, dtype=torch.uint8, device="cuda")
    d_act = torch.zeros(B, dtype=torch.int32, device="cuda")
    worst = 0.0
    for t in range(steps):
        obs = _obs(rng, B)
        d_obs = _dev(obs)
        pb.forward(d_obs.data_ptr(), B, d_probs.data_ptr(), d_value.data_ptr())
        pb.synchronize()
        probs, value, h = policy_ref.forward_batched(params, obs, h, a)
        np.testing.assert_allclose(d_probs.cpu().numpy(), probs, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(d_value.cpu().numpy(), value, rtol=RTOL, atol=ATOL)
        worst = max(worst, float(np.max(np.abs(d_probs.cpu().numpy() - probs) / probs)))
        if B <= 37:
            hg, _ = _memory(pb, B)
            np.testing.assert_allclose(hg, h, rtol=RTOL, atol=ATOL * 10)
        # greedy action -> one-hot memory, command chars
        pb.act(d_probs.data_ptr(), B, d_cmd.data_ptr(), greedy=True, d_action_ptr=d_act.data_ptr())
        pb.synchronize()
        acts = d_act.cpu().numpy()
        v = policy_ref.action_weights(d_probs.cpu().numpy())
        assert (acts == v.argmax(axis=1)).all()
        assert bytes(d_cmd.cpu().numpy().tolist()) == "".join(policy.ACTION_STRING[i] for i in acts).encode()
        a = np.eye(9, dtype=np.float32)[acts]
        if B <= 37:
            _, ag = _memory(pb, B)
            assert (ag == a).all()
    print("B=%d worst relative error on probabilities %.3g" % (B, worst))


def test_reset_memory_mask():
    B = 6
    params = policy.init_parameters(seed=3)
    pb = policy.PolicyBatch(params, B)
    rng = np.random.default_rng(0)
    for b in range(B):
        pb.set_memory(b, rng.normal(size=(2, 160)), np.eye(9)[3])
    mask = _dev(np.array([1, 0, 0, 1, 0, 1], dtype=np.uint8))
    pb.reset_memory(mask.data_ptr())
    pb.synchronize()
    for b in range(B):
        h, a = pb.get_memory(b)
        if b in (0, 3, 5):
            assert not h.any() and (a == np.eye(9)[0]).all()
        else:
            assert h.any() and (a == np.eye(9)[3]).all()
    pb.reset_memory(None)
    h, a = pb.get_memory(1)
    assert not h.any() and a[0] == 1
    # fewer agents than the policy holds, with a mask of just that many bytes: the tail is neither read nor reset
    for b in range(B):
        pb.set_memory(b, rng.normal(size=(2, 160)), np.eye(9)[3])
    short = _dev(np.array([1, 1, 0], dtype=np.uint8))
    pb.reset_memory(short.data_ptr(), agents=3)
    pb.synchronize()
    for b in range(B):
        h, a = pb.get_memory(b)
        assert (not h.any()) == (b in (0, 1)), b


def test_sampling_follows_the_predict_distribution():
    """Agent.hpp:204-211: weights v (v[0] = 0.5, the rest rescaled) through discrete_distribution."""
    B = 20000
    params = policy.init_parameters(seed=1)
    pb = policy.PolicyBatch(params, B)
    p = np.array([0.3, 0.05, 0.15, 0.1, 0.05, 0.05, 0.1, 0.1, 0.1], dtype=np.float32)
    d_probs = _dev(np.tile(p, (B, 1)))
    d_cmd = torch.zeros(B, dtype=torch.uint8, device="cuda")
    d_act = torch.zeros(B, dtype=torch.int32, device="cuda")
    counts = np.zeros(9)
    for draw in range(5):
        pb.act(d_probs.data_ptr(), B, d_cmd.data_ptr(), seed=77, d_action_ptr=d_act.data_ptr())
        pb.synchronize()
        acts = d_act.cpu().numpy()
        if draw == 0:
            first = acts.copy()
        counts += np.bincount(acts, minlength=9)
    v = policy_ref.action_weights(p)
    want = v / v.sum()
    got = counts / counts.sum()
    assert np.abs(got - want).max() < 0.006, (got, want)  # ~4 sigma at 100 000 draws
    # the stream is a pure function of (seed, agent, draw index)
    pb2 = policy.PolicyBatch(params, B)
    pb2.act(d_probs.data_ptr(), B, d_cmd.data_ptr(), seed=77, d_action_ptr=d_act.data_ptr())
    pb2.synchronize()
    assert (d_act.cpu().numpy() == first).all()
    pb2.act(d_probs.data_ptr(), B, d_cmd.data_ptr(), seed=78, d_action_ptr=d_act.data_ptr())
    pb2.synchronize()
    assert (d_act.cpu().numpy() != first).any()


def test_closed_loop_with_the_simulator():
    """observe_device -> forward -> act -> step_device with nothing leaving the GPU, shadowed on the CPU by the
    oracle simulator driven with the commands the GPU chose and by the reference network fed the oracle's
    observations: arena state stays bit-identical and the probabilities stay within tolerance at every step."""
    arenas, steps = 48, 12
    w = config.baseline_workload("C2", arenas=arenas)
    o, g = Oracle(w), env.ArenaBatch(w)
    tb, sr = w.seeds()
    o.reset(tb, sr), g.reset(tb, sr)
    B = arenas * w.cfg.n_agents
    params = policy.init_parameters(seed=9)
    pb = policy.PolicyBatch(params, B)
    d_obs = torch.zeros((B, 32, 31, 31), dtype=torch.float32, device="cuda")
    d_probs = torch.zeros((B, 9), dtype=torch.float32, device="cuda")
    d_value = torch.zeros(B, dtype=torch.float32, device="cuda")
    d_cmd = torch.zeros(B, dtype=torch.uint8, device="cuda")
    d_act = torch.zeros(B, dtype=torch.int32, device="cuda")
    h = np.zeros((2, B, 160), dtype=np.float32)
    a = np.zeros((B, 9), dtype=np.float32)
    a[:, 0] = 1
    for t in range(steps):
        g.observe_device(d_obs.data_ptr())
        pb.forward(d_obs.data_ptr(), B, d_probs.data_ptr(), d_value.data_ptr())
        pb.act(d_probs.data_ptr(), B, d_cmd.data_ptr(), seed=5, d_action_ptr=d_act.data_ptr())
        g.step_device(d_cmd.data_ptr(), 1)
        g.synchronize()
        pb.synchronize()
        probs, value, h = policy_ref.forward_batched(params, o.observe().reshape(B, 32, 31, 31), h, a)
        np.testing.assert_allclose(d_probs.cpu().numpy(), probs, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(d_value.cpu().numpy(), value, rtol=RTOL, atol=ATOL)
        acts = d_act.cpu().numpy()
        a = np.eye(9, dtype=np.float32)[acts]
        cmd = d_cmd.cpu().numpy()
        assert bytes(cmd.tolist()) == "".join(policy.ACTION_STRING[i] for i in acts).encode()
        o.step(cmd)
        assert (o.digest() == g.digest()).all()
    assert len(set(acts.tolist())) > 1


def test_forward_rejects_bad_arguments():
    pb = policy.PolicyBatch(policy.init_parameters(0), 4)
    d = torch.zeros(16, device="cuda")
    with pytest.raises(env.StrikeForceError, match="out of range"):
        pb.forward(d.data_ptr(), 5, d.data_ptr(), d.data_ptr())
    with pytest.raises(env.StrikeForceError, match="null"):
        pb.forward(0, 4, d.data_ptr(), d.data_ptr())
    with pytest.raises(env.StrikeForceError, match="9 chars"):
        pb.act(d.data_ptr(), 4, d.data_ptr(), action_string="+x")


@pytest.mark.parametrize("M,N,K", [(1, 160, 32), (33, 160, 160), (4096, 480, 160), (20000, 160, 352), (70001, 320, 288),
                                   (70001, 160, 288), (40000, 160, 1440), (5001, 160, 1440)])
def test_gemm_kernel_matches_torch(M, N, K):
    """The MFMA GEMM alone (both block shapes; one run per tile and the balanced stream-K split with tiles cut
    across two and three runs; ragged M, several N tiles) vs a float64 matmul on the CPU: |err| <= 2e-6 * sum|a*b| (f32 fmaf chain, guide: ~1e-7 * sum|a*b| typical)."""
    pb = policy.PolicyBatch(policy.init_parameters(0), 4)
    rng = np.random.default_rng(M)
    a = rng.normal(size=(M, K)).astype(np.float32)
    w = rng.normal(size=(N, K)).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    da, dw, db = _dev(a), _dev(w), _dev(bias)
    dc = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    pb.gemm(da.data_ptr(), K, dw.data_ptr(), db.data_ptr(), dc.data_ptr(), N, M, N, K)
    pb.synchronize()
    want = a.astype(np.float64) @ w.astype(np.float64).T + bias
    bound = 2e-6 * (np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T) + 1e-6
    assert (np.abs(dc.cpu().numpy() - want) <= bound).all()


@pytest.mark.parametrize("M,N,K", [(1, 160, 32), (300, 160, 160), (20000, 160, 352), (70001, 320, 288),
                                   (70001, 160, 288), (200704, 160, 1440), (36864, 160, 1440), (5001, 160, 1440)])
def test_split_gemm_kernel_matches_float64(M, N, K):
    """k_gemm_b3 alone (every f32 operand as three bf16 parts, six bf16 MFMAs per block: conv1 / conv2's kernel) vs a
    float64 matmul on the CPU, same bound as the f32 kernel: |err| <= 2e-6 * sum|a*b|.  One run per tile and the stream-K
    split, ragged M, two column strips; operands spanning 2^+-20 so that all three parts of a number matter."""
    pb = policy.PolicyBatch(policy.init_parameters(0), 4)
    rng = np.random.default_rng(M + 1)
    a = (rng.normal(size=(M, K)) * np.exp2(rng.integers(-20, 20, size=(M, 1)))).astype(np.float32)
    w = (rng.normal(size=(N, K)) * np.exp2(rng.integers(-20, 20, size=(N, 1)))).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    da, dw, db = _dev(a), _dev(w), _dev(bias)
    dc = torch.zeros((M, N), dtype=torch.float32, device="cuda")
    pb.gemm_split(da.data_ptr(), K, dw.data_ptr(), db.data_ptr(), dc.data_ptr(), N, M, N, K)
    pb.synchronize()
    got = dc.cpu().numpy()
    rows = slice(None) if M <= 70001 else rng.choice(M, size=20000, replace=False)
    a, got = a[rows], got[rows]
    want = a.astype(np.float64) @ w.astype(np.float64).T + bias
    mag = np.abs(a).astype(np.float64) @ np.abs(w).astype(np.float64).T + np.abs(bias)  # sum|a*b| + |bias|
    err = np.abs(got - want)
    print("worst |err| / (sum|a*b| + |bias|) = %.3g" % float(np.max(err / mag)))
    assert (err <= 2e-6 * mag).all()


def test_split_and_f32_convolutions_agree():
    """The network with conv1 / conv2 on the bf16-split kernel (default) and on the f32 kernel (SF_POLICY_F32_CONV=1):
    same probabilities and value within the tolerance of this file."""
    rng = np.random.default_rng(5)
    params = policy.init_parameters(seed=9)
    B = 2000  # conv1 M = 98 000, conv2 M = 18 000: both on the split kernel
    obs = _dev(_obs(rng, 50))
    d_obs = obs[torch.arange(B, device="cuda") % 50].contiguous()
    out = []
    for f32 in ("0", "1"):
        os.environ["SF_POLICY_F32_CONV"] = f32
        os.environ["SF_POLICY_LAYERED"] = "1"
        try:
            pb = policy.PolicyBatch(params, B)
        finally:
            del os.environ["SF_POLICY_F32_CONV"]
            del os.environ["SF_POLICY_LAYERED"]
        d_probs = torch.zeros((B, 9), dtype=torch.float32, device="cuda")
        d_value = torch.zeros(B, dtype=torch.float32, device="cuda")
        for _ in range(2):
            pb.forward(d_obs.data_ptr(), B, d_probs.data_ptr(), d_value.data_ptr())
        pb.synchronize()
        out.append((d_probs.cpu().numpy(), d_value.cpu().numpy()))
        pb.close()
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=RTOL, atol=ATOL)
    assert not np.array_equal(out[0][0], out[1][0])  # two different kernels did run


def test_the_convolution_stack_alone_matches_float64(cnn):
    """GameCNN::forward alone (sf_policy_features): the 160 features behind the four bias-free convolutions against the
    same chain in float64, on sparse inputs (1 % non-zero, what an observation is), dense ones (30 %) and single
    non-zeros in the window's corners, on its edges and inside (one row of the composed matrix each: every boundary
    case of the 3x3 / stride-2 windows).  Two gates: |err| <= 1e-8 * (the chain applied to |x| with |W|: what the
    roundings of a four-layer f32 evaluation are proportional to; measured 2e-10 composed, 2e-9 layered), and |err| <=
    c * the agent's largest feature, c = 2e-6 for the composed matrix (its entries are within half an ulp of the exact
    composition; ~20-term f32 partial sums, combined in f64) and 1e-5 for the four layers one after the other."""
    import torch.nn.functional as F
    rng = np.random.default_rng(77)
    params = policy.init_parameters(seed=31)
    B = 48
    x = np.zeros((B, 32, 31, 31), dtype=np.float32)
    x[:32] = rng.uniform(-2.0, 2.0, size=(32, 32, 31, 31)) * (rng.uniform(size=(32, 32, 31, 31)) < 0.01)
    x[32:40] = rng.uniform(-2.0, 2.0, size=(8, 32, 31, 31)) * (rng.uniform(size=(8, 32, 31, 31)) < 0.3)
    for b, (ch, y, xx) in enumerate([(0, 0, 0), (31, 30, 30), (5, 0, 30), (7, 30, 0), (3, 15, 15), (9, 1, 2), (10, 2, 1), (11, 29, 28)]):
        x[40 + b, ch, y, xx] = 1.5
    want = torch.from_numpy(x).double()
    mag = want.abs()
    for i in range(4):
        w = torch.from_numpy(params["backbone.cnn.conv%d.weight" % i]).double()
        want, mag = F.conv2d(want, w, stride=2), F.conv2d(mag, w.abs(), stride=2)
    want, mag = want.reshape(B, 160).numpy(), mag.reshape(B, 160).numpy()
    pb = policy.PolicyBatch(params, B)
    d_obs = _dev(x)
    d_feat = torch.zeros((B, 160), dtype=torch.float32, device="cuda")
    pb.features(d_obs.data_ptr(), B, d_feat.data_ptr())
    pb.synchronize()
    got = d_feat.cpu().numpy().astype(np.float64)
    pb.close()
    err = np.abs(got - want)
    scale = np.abs(want).max(axis=1, keepdims=True)  # an agent's largest feature
    print("%s: worst |err| / magnitude %.3g, / the agent's largest feature %.3g" %
          (cnn, float(np.max(err / np.maximum(mag, 1e-30))), float(np.max(err / scale))))
    assert (err <= 1e-8 * mag + 1e-30).all()
    assert (err <= (2e-6 if cnn == "folded" else 1e-5) * scale).all()
    assert np.abs(want[40:]).max() > 1e-4  # the single non-zeros do reach the features


def test_folded_and_layered_convolutions_agree():
    """The convolution stack as one composed matrix on the non-zeros (default) and layer by layer (SF_POLICY_LAYERED=1) on
    real observations of the simulator, over recurrent steps: same probabilities, value and state within this file's
    tolerance; and the folded form's features alone against float64 of the composed map."""
    A = 64
    w = config.baseline_workload("C3", arenas=A)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    n = A * w.cfg.n_agents
    cmds, _ = config.bench_commands(A, w.cfg.n_agents, 120)
    params = policy.init_parameters(seed=21)
    pbs = []
    for layered in ("0", "1"):
        os.environ["SF_POLICY_LAYERED"] = layered
        try:
            pbs.append(policy.PolicyBatch(params, n))
        finally:
            del os.environ["SF_POLICY_LAYERED"]
    d_obs = torch.zeros((n, 32, 31, 31), dtype=torch.float32, device="cuda")
    outs = [(torch.zeros((n, 9), dtype=torch.float32, device="cuda"), torch.zeros(n, dtype=torch.float32, device="cuda")) for _ in pbs]
    differ = False
    for t in range(6):
        for s in range(t * 20, (t + 1) * 20):
            g.step(cmds[s])
        g.observe_device(d_obs.data_ptr())
        g.synchronize()
        for pb, (dp, dv) in zip(pbs, outs):
            pb.forward(d_obs.data_ptr(), n, dp.data_ptr(), dv.data_ptr())
            pb.synchronize()
        np.testing.assert_allclose(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy(), rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy(), rtol=RTOL, atol=ATOL)
        for b in (0, n // 2, n - 1):
            np.testing.assert_allclose(pbs[0].get_memory(b)[0], pbs[1].get_memory(b)[0], rtol=RTOL, atol=ATOL * 10)
        differ = differ or not np.array_equal(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy())
    assert differ  # two different evaluations did run
    for pb in pbs:
        pb.close()
    g.close()


def test_hip_path_reproduces_the_committed_vectors(cnn):
    """tests/golden/policy_vectors.json = outputs of the REFERENCE's AgentModel (bots/bot-0.5/Modules.hpp:26-180
    compiled unedited, oracle/ref_modules.py; generator tests/golden/make_policy_vectors.py): the GPU simulator's
    observations of the same seeded run through the HIP network give the committed probabilities, values and state
    checksums."""
    import json
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "policy_vectors.json")))
    assert "libsf_refmodules" in want["_generator"]
    A = want["arenas"]
    w = config.baseline_workload(want["workload"], arenas=A)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    between = want["sim_steps_between"]
    cmds, _ = config.bench_commands(A, w.cfg.n_agents, len(want["steps"]) * between)
    pb = policy.PolicyBatch(policy.init_parameters(seed=want["param_seed"]), A)
    d_obs = torch.zeros((A, 32, 31, 31), dtype=torch.float32, device="cuda")
    d_probs = torch.zeros((A, 9), dtype=torch.float32, device="cuda")
    d_value = torch.zeros(A, dtype=torch.float32, device="cuda")
    d_cmd = torch.zeros(A, dtype=torch.uint8, device="cuda")
    d_act = torch.zeros(A, dtype=torch.int32, device="cuda")
    for t, step in enumerate(want["steps"]):
        for s in range(t * between, (t + 1) * between):
            g.step(cmds[s])
        g.observe_device(d_obs.data_ptr())
        pb.forward(d_obs.data_ptr(), A, d_probs.data_ptr(), d_value.data_ptr())
        # feed back the arg-max of the raw probabilities, as the generator did
        acts = d_probs.cpu().numpy().argmax(axis=1)
        assert acts.tolist() == step["action_fed_back"]
        for b in range(A):
            h, _ = pb.get_memory(b)
            pb.set_memory(b, h, np.eye(9, dtype=np.float32)[acts[b]])
        g.synchronize()
        assert int((d_obs != 0).sum().item()) == step["obs_nonzero"]
        np.testing.assert_allclose(d_probs.cpu().numpy(), step["probs"], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(d_value.cpu().numpy(), step["value"], rtol=RTOL, atol=ATOL)
        hs = [float(np.abs(np.stack([pb.get_memory(b)[0][gi] for b in range(A)])).sum()) for gi in range(2)]
        np.testing.assert_allclose(hs, step["h_abs_sum"], rtol=RTOL)


def test_closed_loop_resets_memory_when_a_game_restarts():
    """Timer-mode arenas with a 6-frame limit restart every few steps (auto_reset): sf_done_device marks the agents
    whose game just restarted and sf_policy_reset_memory gives them the memory of a new Agent, which is what the
    reference does per game (prepare(), gameplay.hpp:481).  Shadowed by oracle simulator + reference network."""
    from strikeforce_amd import abi
    arenas, steps = 6, 14
    cfg = config.make_config(arenas, 32, 32, H=1, Z=4, B=16, P=4, mode=abi.MODE_TIMER, n_agents=1, auto_reset=1,
                             timer_frames=6)
    m, pmap = config.synthetic_map(32, 32)
    w = config.Workload("timer6", cfg, m, pmap)
    o, g = Oracle(w), env.ArenaBatch(w)
    tb, sr = w.seeds()
    o.reset(tb, sr), g.reset(tb, sr)
    B = arenas
    params = policy.init_parameters(seed=4)
    pb = policy.PolicyBatch(params, B)
    d_obs = torch.zeros((B, 32, 31, 31), dtype=torch.float32, device="cuda")
    d_probs = torch.zeros((B, 9), dtype=torch.float32, device="cuda")
    d_value = torch.zeros(B, dtype=torch.float32, device="cuda")
    d_cmd = torch.zeros(B, dtype=torch.uint8, device="cuda")
    d_act = torch.zeros(B, dtype=torch.int32, device="cuda")
    d_new = torch.zeros(B, dtype=torch.uint8, device="cuda")
    h = np.zeros((2, B, 160), dtype=np.float32)
    a = np.eye(9, dtype=np.float32)[[0] * B]
    restarts = 0
    for t in range(steps):
        g.observe_device(d_obs.data_ptr())
        pb.forward(d_obs.data_ptr(), B, d_probs.data_ptr(), d_value.data_ptr())
        pb.act(d_probs.data_ptr(), B, d_cmd.data_ptr(), seed=2, d_action_ptr=d_act.data_ptr())
        g.step_device(d_cmd.data_ptr(), 1)
        g.done_device(d_new.data_ptr())
        pb.reset_memory(d_new.data_ptr())
        g.synchronize(), pb.synchronize()
        probs, value, h = policy_ref.forward_batched(params, o.observe().reshape(B, 32, 31, 31), h, a)
        np.testing.assert_allclose(d_probs.cpu().numpy(), probs, rtol=RTOL, atol=ATOL)
        a = np.eye(9, dtype=np.float32)[d_act.cpu().numpy()]
        o.step(d_cmd.cpu().numpy())
        assert (o.digest() == g.digest()).all()
        new = o.done()  # per (arena, agent): one agent per arena here
        assert (d_new.cpu().numpy() == new).all()
        for b in np.nonzero(new)[0]:  # reset_memory(): h_state = 0, action_input = one-hot(0)   Modules.hpp:95-100
            h[:, b] = 0
            a[b] = np.eye(9, dtype=np.float32)[0]
        restarts += int(new.sum())
    assert restarts >= arenas  # every arena restarted at least once
    hg, ag = pb.get_memory(0)
    np.testing.assert_allclose(hg, h[:, 0], rtol=RTOL, atol=ATOL * 10)


def test_large_batch_is_consistent_with_small_batch():
    """18 000 agents (2.2 GB of observations, conv0 with 4 M rows: byte offsets beyond 2^31, thousands of stream-K
    tiles) fed 40 distinct observations in rotation give, row for row, what a 40-agent batch gives."""
    rng = np.random.default_rng(77)
    params = policy.init_parameters(seed=6)
    small, big = 40, 18000
    obs = _obs(rng, small)
    d_small = _dev(obs)
    idx = torch.arange(big, device="cuda") % small
    d_big = d_small[idx].contiguous()
    out = {}
    for n, d in ((small, d_small), (big, d_big)):
        pb = policy.PolicyBatch(params, n)
        d_probs = torch.zeros((n, 9), dtype=torch.float32, device="cuda")
        d_value = torch.zeros(n, dtype=torch.float32, device="cuda")
        for _ in range(2):  # two recurrent steps
            pb.forward(d.data_ptr(), n, d_probs.data_ptr(), d_value.data_ptr())
        pb.synchronize()
        out[n] = (d_probs.cpu().numpy(), d_value.cpu().numpy())
        pb.close()
    ps, vs = out[small]
    pbig, vbig = out[big]
    ref = np.arange(big) % small
    # different block shapes / K splits for the two batch sizes: equal up to summation order
    np.testing.assert_allclose(pbig, ps[ref], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(vbig, vs[ref], rtol=RTOL, atol=ATOL)


def test_closed_loop_multi_agent_arenas():
    """BASELINE configs[4] (Battle Royale, 8 commanded humans per arena): the policy's agent index is
    arena * n_agents + agent, the order of the observation buffer and of the command array; shadowed by the oracle."""
    arenas, steps = 2, 6
    w = config.baseline_workload("C5", arenas=arenas)
    o, g = Oracle(w), env.ArenaBatch(w)
    tb, sr = w.seeds()
    o.reset(tb, sr), g.reset(tb, sr)
    B = arenas * w.cfg.n_agents
    assert w.cfg.n_agents == 8
    params = policy.init_parameters(seed=12)
    pb = policy.PolicyBatch(params, B)
    d_obs = torch.zeros((B, 32, 31, 31), dtype=torch.float32, device="cuda")
    d_probs = torch.zeros((B, 9), dtype=torch.float32, device="cuda")
    d_value = torch.zeros(B, dtype=torch.float32, device="cuda")
    d_cmd = torch.zeros(B, dtype=torch.uint8, device="cuda")
    d_act = torch.zeros(B, dtype=torch.int32, device="cuda")
    h = np.zeros((2, B, 160), dtype=np.float32)
    a = np.eye(9, dtype=np.float32)[[0] * B]
    for t in range(steps):
        g.observe_device(d_obs.data_ptr())
        pb.forward(d_obs.data_ptr(), B, d_probs.data_ptr(), d_value.data_ptr())
        pb.act(d_probs.data_ptr(), B, d_cmd.data_ptr(), seed=3, d_action_ptr=d_act.data_ptr())
        g.step_device(d_cmd.data_ptr(), 1)
        g.synchronize(), pb.synchronize()
        probs, value, h = policy_ref.forward_batched(params, o.observe().reshape(B, 32, 31, 31), h, a)
        np.testing.assert_allclose(d_probs.cpu().numpy(), probs, rtol=RTOL, atol=ATOL)
        a = np.eye(9, dtype=np.float32)[d_act.cpu().numpy()]
        o.step(d_cmd.cpu().numpy())
        assert (o.digest() == g.digest()).all()
    assert np.abs(probs[0] - probs[1]).max() > 1e-4  # different agents see different windows
