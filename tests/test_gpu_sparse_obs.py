"""The observation handed from the simulator to the bot network as the list of its non-zero floats
(sf_observe_sparse_device -> sf_policy_forward_sparse, include/strikeforce*.h): the list is exactly the dense observation's
non-zeros in the dense buffer's order, and the network's outputs are bit-identical to the dense pair of calls."""
import numpy as np
import pytest
import torch

from oracle_lib import Oracle
from strikeforce_amd import config, env, policy

pytestmark = pytest.mark.gpu
CAP = 2048


def _bufs(B):
    return (torch.zeros((B, CAP), dtype=torch.int32, device="cuda"), torch.zeros((B, CAP), dtype=torch.float32, device="cuda"),
            torch.zeros(B, dtype=torch.int32, device="cuda"), torch.zeros((B, 160), dtype=torch.float32, device="cuda"))


def _advance(w, g, steps):
    cmds, _ = config.bench_commands(w.cfg.arenas, w.cfg.n_agents, steps)
    d = torch.from_numpy(np.ascontiguousarray(cmds)).cuda()
    torch.cuda.synchronize()
    g.step_device(d.data_ptr(), steps)
    g.synchronize()


@pytest.mark.parametrize("which", ["C2", "C3", "C5", "KITS"])
def test_the_list_is_the_dense_observation(which):
    w = config.baseline_workload(which, arenas=6)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    B = w.cfg.arenas * w.cfg.n_agents
    _advance(w, g, 150)  # zombies, bullets, chests and corpses in view
    d_obs = torch.zeros((B, 32, 31, 31), dtype=torch.float32, device="cuda")
    keys, vals, counts, pov = _bufs(B)
    g.observe_device(d_obs.data_ptr())
    g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP)
    g.synchronize()
    obs = d_obs.cpu().numpy().reshape(B, -1)
    k, v, n, pv = keys.cpu().numpy().view(np.uint32), vals.cpu().numpy(), counts.cpu().numpy().view(np.uint32), pov.cpu().numpy()
    assert n.max() <= CAP and n.max() > 0
    for a in range(B):
        nz = np.flatnonzero(obs[a])           # ascending dense index = channel, row, column
        assert n[a] == len(nz)
        ch, r = np.divmod(nz, 961)
        y, x = np.divmod(r, 31)
        assert np.array_equal(k[a, :n[a]], (ch * 9) | (y << 9) | (x << 14))
        assert np.array_equal(v[a, :n[a]].view(np.uint32), obs[a][nz].view(np.uint32))
    o4 = obs.reshape(B, 32, 31, 31)
    cells = [(14, 15), (15, 14), (15, 15), (15, 16), (16, 15)]  # Modules.hpp:114-121
    want = np.stack([o4[:, :, yy, xx] for yy, xx in cells], axis=1).reshape(B, 160)
    assert np.array_equal(pv.view(np.uint32), want.view(np.uint32))


def test_the_list_is_the_dense_observation_with_large_pools():
    """Zombie and exit tables of more than 64 slots (in LDS for the step kernels, read where they lie by both observation
    kernels): 4600 steps into tests/test_large_pools.py's world, ~140 zombies in three 64-slot words and ~100 exits.  Every
    agent's list equals its dense observation; a window with more than 48 occupied cells says "crowded" instead."""
    import test_large_pools
    w = test_large_pools.world(64, 12)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    B = w.cfg.arenas
    cmds, _ = config.bench_commands(B, 1, 4600, seed0=4321)
    d = torch.from_numpy(np.ascontiguousarray(cmds)).cuda()
    g.step_device(d.data_ptr(), 4600)
    d_obs = torch.zeros((B, 32, 31, 31), dtype=torch.float32, device="cuda")
    keys, vals, counts, pov = _bufs(B)
    g.observe_device(d_obs.data_ptr())
    g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP)
    g.synchronize()
    assert max(sum(z.alive for z in g.dump(a).zombies) for a in range(B)) > 128
    obs = d_obs.cpu().numpy().reshape(B, -1)
    k, v, n = keys.cpu().numpy().view(np.uint32), vals.cpu().numpy(), counts.cpu().numpy().view(np.uint32)
    compared = 0
    for a in range(B):
        nz = np.flatnonzero(obs[a])
        if n[a] == 0xFFFFFFFF:
            assert len(np.unique(nz % 961)) > 48  # (more occupied cells than the list kernel has records)
            continue
        assert n[a] == len(nz) and n[a] > 100
        ch, r = np.divmod(nz, 961)
        y, x = np.divmod(r, 31)
        assert np.array_equal(k[a, :n[a]], (ch * 9) | (y << 9) | (x << 14))
        assert np.array_equal(v[a, :n[a]].view(np.uint32), obs[a][nz].view(np.uint32))
        compared += 1
    assert compared >= B // 2


def test_sparse_and_dense_forward_are_bit_identical(cnn):
    w = config.baseline_workload("C3", arenas=40)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    B = w.cfg.arenas * w.cfg.n_agents
    params = policy.init_parameters(seed=4)
    dense, sparse = policy.PolicyBatch(params, B), policy.PolicyBatch(params, B)
    d_obs = torch.zeros((B, 32, 31, 31), dtype=torch.float32, device="cuda")
    keys, vals, counts, pov = _bufs(B)
    out = [(torch.zeros((B, 9), device="cuda"), torch.zeros(B, device="cuda")) for _ in range(2)]
    for _ in range(4):  # recurrent steps, the world moving in between
        _advance(w, g, 25)
        g.observe_device(d_obs.data_ptr())
        g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP)
        dense.forward(d_obs.data_ptr(), B, out[0][0].data_ptr(), out[0][1].data_ptr())
        sparse.forward_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP, B, out[1][0].data_ptr(),
                              out[1][1].data_ptr())
        dense.synchronize(), sparse.synchronize()
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert sparse.sparse_overflows() == 0
    assert float(out[0][0].std()) > 0


def test_a_list_that_does_not_fit_is_counted(cnn):
    w = config.baseline_workload("C2", arenas=8)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    B = w.cfg.arenas * w.cfg.n_agents
    pb = policy.PolicyBatch(policy.init_parameters(0), B)
    cap = 16  # far too small: every agent sees more than 16 non-zero floats
    keys = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    vals = torch.zeros((B, cap), dtype=torch.float32, device="cuda")
    counts, pov = torch.zeros(B, dtype=torch.int32, device="cuda"), torch.zeros((B, 160), device="cuda")
    probs, value = torch.zeros((B, 9), device="cuda"), torch.zeros(B, device="cuda")
    g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), cap)
    g.synchronize()
    assert int(counts.min()) > cap  # the true counts are reported, the lists cut
    pb.forward_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), cap, B, probs.data_ptr(), value.data_ptr())
    assert pb.sparse_overflows() == B
    assert pb.sparse_overflows() == 0  # cleared by the query


def test_a_list_capacity_above_the_kernels_limit_is_refused(cnn):
    """Above SF_POLICY_LIST_MAX (2048) the network's kernels would call an agent with 2048 < count <= cap "overflowed"
    while sf_observe_overflow_device, which only knows cap, never wrote its dense row: both entry points refuse such a
    cap (SF_ERR_ARG), so "the list did not fit" means count > cap (or the marker) on both sides."""
    B, cap = 4, 4096
    pb = policy.PolicyBatch(policy.init_parameters(0), B)
    keys = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    vals = torch.zeros((B, cap), dtype=torch.float32, device="cuda")
    counts, pov = torch.zeros(B, dtype=torch.int32, device="cuda"), torch.zeros((B, 160), device="cuda")
    probs, value = torch.zeros((B, 9), device="cuda"), torch.zeros(B, device="cuda")
    dense = torch.full((B, 32 * 31 * 31), float("nan"), device="cuda")
    for d in (None, dense.data_ptr()):
        with pytest.raises(env.StrikeForceError, match="SF_POLICY_LIST_MAX"):
            pb.forward_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), cap, B, probs.data_ptr(),
                              value.data_ptr(), d_dense_ptr=d)
    pb.forward_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), 2048, B, probs.data_ptr(), value.data_ptr())
    pb.synchronize()


def test_crowded_windows_are_marked_not_truncated():
    """MAXCAP (64 humans, 64 zombies, 256 bullet slots on 48 x 48): windows with more occupied cells than the kernel has
    records take the dense call's slow path; the list form must say so (count 0xffffffff) instead of handing over a partial
    list, and every other agent's list must still equal its dense observation."""
    w = config.baseline_workload("MAXCAP", arenas=4)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    B = w.cfg.arenas * w.cfg.n_agents
    marked = 0
    for _ in range(6):
        _advance(w, g, 40)
        d_obs = torch.zeros((B, 32, 31, 31), dtype=torch.float32, device="cuda")
        keys, vals, counts, pov = _bufs(B)
        g.observe_device(d_obs.data_ptr())
        g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP)
        g.synchronize()
        obs = d_obs.cpu().numpy().reshape(B, -1)
        k, v, n = keys.cpu().numpy().view(np.uint32), vals.cpu().numpy(), counts.cpu().numpy().view(np.uint32)
        for a in range(B):
            nz = np.flatnonzero(obs[a])
            if n[a] == 0xFFFFFFFF:
                marked += 1
                continue
            assert n[a] == len(nz)
            m = min(len(nz), CAP)
            ch, r = np.divmod(nz[:m], 961)
            y, x = np.divmod(r, 31)
            assert np.array_equal(k[a, :m], (ch * 9) | (y << 9) | (x << 14))
            assert np.array_equal(v[a, :m].view(np.uint32), obs[a][nz[:m]].view(np.uint32))
    print("agents marked as crowded:", marked)


@pytest.mark.parametrize("which,cap,arenas", [("C2", 16, 8), ("C3", 200, 24), ("MAXCAP", 2048, 4)])
def test_agents_whose_list_does_not_fit_are_evaluated_from_the_dense_fallback(which, cap, arenas, cnn):
    """No agent is ever evaluated on a blank window: with sf_observe_overflow_device + sf_policy_forward_sparse_or_dense
    the agents whose list did not fit — a cap far too small (16: every agent), a cap that cuts some lists (200), the
    crowded-window marker (MAXCAP) — are redone from their dense observation on the device, no host round trip.  The
    outputs equal the dense pair of calls bit for bit over recurrent steps, for every agent; nothing is counted."""
    w = config.baseline_workload(which, arenas=arenas)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    B = w.cfg.arenas * w.cfg.n_agents
    params = policy.init_parameters(seed=4)
    dense, sparse = policy.PolicyBatch(params, B), policy.PolicyBatch(params, B)
    d_obs = torch.zeros((B, 32, 31, 31), dtype=torch.float32, device="cuda")
    d_fb = torch.full((B, 32, 31, 31), float("nan"), dtype=torch.float32, device="cuda")  # rows nobody writes must not be read
    keys = torch.zeros((B, cap), dtype=torch.int32, device="cuda")
    vals = torch.zeros((B, cap), dtype=torch.float32, device="cuda")
    counts, pov = torch.zeros(B, dtype=torch.int32, device="cuda"), torch.zeros((B, 160), device="cuda")
    out = [(torch.zeros((B, 9), device="cuda"), torch.zeros(B, device="cuda")) for _ in range(2)]
    flagged = 0
    for _ in range(4):
        _advance(w, g, 40)
        g.observe_device(d_obs.data_ptr())
        dense.forward(d_obs.data_ptr(), B, out[0][0].data_ptr(), out[0][1].data_ptr())
        g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), cap)
        g.observe_overflow_device(counts.data_ptr(), cap, d_fb.data_ptr(), pov.data_ptr())
        sparse.forward_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), cap, B,
                              out[1][0].data_ptr(), out[1][1].data_ptr(), d_dense_ptr=d_fb.data_ptr())
        g.synchronize(), dense.synchronize(), sparse.synchronize()
        n = counts.cpu().numpy().view(np.uint32)
        flagged += int(((n > cap) | (n == 0xFFFFFFFF)).sum())
        assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
        for b in range(B):
            hd, ad = dense.get_memory(b)
            hs, as_ = sparse.get_memory(b)
            assert np.array_equal(hd, hs)
    assert sparse.sparse_overflows() == 0
    if which != "MAXCAP":
        assert flagged > 0


@pytest.mark.parametrize("resets", ["mask", "words", "both"])
def test_predict_is_reset_forward_act_in_one_call(resets, cnn):
    """sf_policy_predict_sparse = sf_policy_reset_memory + sf_policy_forward_sparse_or_dense + sf_policy_act folded into the
    forward's two launches (Agent::predict + update are one call in the reference, Agent.hpp:200-222): probabilities,
    value, command, action, and every agent's memory afterwards equal the three calls' bit for bit over recurrent steps —
    with the restart flags as a byte per agent, as words read in place (sf_done_view_device's layout: one word per arena,
    a stride apart), or both."""
    w = config.baseline_workload("C5", arenas=5)  # 8 agents per arena
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    B, G = w.cfg.arenas * w.cfg.n_agents, w.cfg.n_agents
    params = policy.init_parameters(seed=6)
    three, one = policy.PolicyBatch(params, B), policy.PolicyBatch(params, B)
    keys, vals, counts, pov = _bufs(B)
    d_fb = torch.full((B, 32, 31, 31), float("nan"), dtype=torch.float32, device="cuda")
    out = [(torch.zeros((B, 9), device="cuda"), torch.zeros(B, device="cuda"), torch.zeros(B, dtype=torch.uint8, device="cuda"),
            torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in range(2)]
    rng = np.random.default_rng(11)
    STRIDE = 3
    for step in range(6):
        _advance(w, g, 25)
        mask = (rng.random(B) < 0.3).astype(np.uint8) if resets in ("mask", "both") else np.zeros(B, dtype=np.uint8)
        words = np.zeros(w.cfg.arenas * STRIDE, dtype=np.int32)
        if resets in ("words", "both"):
            words[::STRIDE] = (rng.random(w.cfg.arenas) < 0.4) * rng.integers(1, 300, w.cfg.arenas)
            words[1::STRIDE] = 7  # (the words in between are not flags)
        both = mask | (np.repeat(words[::STRIDE], G) != 0).astype(np.uint8)
        d_mask, d_words, d_both = torch.from_numpy(mask).cuda(), torch.from_numpy(words).cuda(), torch.from_numpy(both).cuda()
        g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP)
        g.observe_overflow_device(counts.data_ptr(), CAP, d_fb.data_ptr(), pov.data_ptr())
        g.synchronize()
        if step:  # (the first step: both start from sf_policy_create's fresh memory)
            three.reset_memory(d_both.data_ptr())
        three.forward_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP, B, out[0][0].data_ptr(),
                             out[0][1].data_ptr(), d_dense_ptr=d_fb.data_ptr())
        three.act(out[0][0].data_ptr(), B, out[0][2].data_ptr(), seed=5, greedy=(step == 4), d_action_ptr=out[0][3].data_ptr())
        one.predict_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP, B, out[1][0].data_ptr(),
                           out[1][1].data_ptr(), out[1][2].data_ptr(), seed=5, greedy=(step == 4), d_action_ptr=out[1][3].data_ptr(),
                           d_dense_ptr=d_fb.data_ptr(), d_reset_mask_ptr=d_mask.data_ptr() if step and resets != "words" else None,
                           reset_words=(d_words.data_ptr(), STRIDE, G) if step and resets != "mask" else None)
        three.synchronize(), one.synchronize()
        for x, y in zip(out[0], out[1]):
            assert torch.equal(x, y)
        for b in range(B):
            h3, a3 = three.get_memory(b)
            h1, a1 = one.get_memory(b)
            assert np.array_equal(h3, h1) and np.array_equal(a3, a1)
        if step != 4:  # (the arg-max of v is always action 0: v[0] = 0.5 is half of the mass)
            assert len(set(out[0][3].cpu().numpy().tolist())) > 1


def test_the_done_view_is_what_sf_done_device_copies():
    """sf_done_view_device: sf_policy_predict_sparse reading the environment's restart flags in place gives what
    sf_done_device + sf_policy_reset_memory give, on games that really end (configs[0], 64 arenas, 4 800 steps)."""
    w = config.baseline_workload("C1", arenas=64)
    g = env.ArenaBatch(w)
    g.reset(*w.seeds())
    B = w.cfg.arenas * w.cfg.n_agents
    view = g.done_view_device()
    assert view[2] == w.cfg.n_agents and view[1] >= 1
    params = policy.init_parameters(seed=8)
    three, one = policy.PolicyBatch(params, B), policy.PolicyBatch(params, B)
    keys, vals, counts, pov = _bufs(B)
    out = [(torch.zeros((B, 9), device="cuda"), torch.zeros(B, device="cuda"), torch.zeros(B, dtype=torch.uint8, device="cuda")) for _ in range(2)]
    d_new = torch.zeros(B, dtype=torch.uint8, device="cuda")
    seen = 0
    for _ in range(12):
        _advance(w, g, 400)
        g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP)
        g.done_device(d_new.data_ptr())
        g.synchronize()
        three.reset_memory(d_new.data_ptr())
        three.forward_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP, B, out[0][0].data_ptr(), out[0][1].data_ptr())
        three.act(out[0][0].data_ptr(), B, out[0][2].data_ptr(), seed=1)
        one.predict_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP, B, out[1][0].data_ptr(),
                           out[1][1].data_ptr(), out[1][2].data_ptr(), seed=1, reset_words=view)
        three.synchronize(), one.synchronize()
        for x, y in zip(out[0], out[1]):
            assert torch.equal(x, y)
        for b in range(B):
            assert np.array_equal(three.get_memory(b)[0], one.get_memory(b)[0])
        seen += int((d_new != 0).sum().item())
    assert seen > 0  # (some games ended)


def test_the_closed_loop_through_predict_equals_the_separate_calls_at_bench_size():
    """bench.py's `policy` loop both ways on configs[2] at 4096 arenas, 150 steps from the same start: six launches per
    step (sf_policy_predict_sparse reading sf_done_view_device's flags) against nine (forward, act, step, sf_done_device,
    sf_policy_reset_memory).  The two worlds stay digest-identical — every command, hence every draw, hence every restart
    was the same — and so do the networks' memories; games end and restart on the way."""
    A, STEPS = 4096, 150
    worlds = []
    for _ in range(2):
        w = config.baseline_workload("C3", arenas=A)
        g = env.ArenaBatch(w)
        g.reset(*w.seeds())
        _advance(w, g, 300)
        worlds.append((w, g, policy.PolicyBatch(policy.init_parameters(seed=0), A), _bufs(A),
                       (torch.zeros((A, 9), device="cuda"), torch.zeros(A, device="cuda"), torch.zeros(A, dtype=torch.uint8, device="cuda"))))
    d_fb = torch.empty((A, 32, 31, 31), dtype=torch.float32, device="cuda")
    d_new = torch.zeros(A, dtype=torch.uint8, device="cuda")
    view = worlds[0][1].done_view_device()
    restarted = 0
    for step in range(STEPS):
        for which, (w, g, pb, (keys, vals, counts, pov), (probs, value, cmd)) in enumerate(worlds):
            g.observe_sparse_device(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP)
            g.observe_overflow_device(counts.data_ptr(), CAP, d_fb.data_ptr(), pov.data_ptr())
            if which == 0:
                pb.predict_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP, A, probs.data_ptr(), value.data_ptr(),
                                  cmd.data_ptr(), seed=3, d_dense_ptr=d_fb.data_ptr(), reset_words=view)
                g.step_device(cmd.data_ptr(), 1)
            else:
                pb.forward_sparse(keys.data_ptr(), vals.data_ptr(), counts.data_ptr(), pov.data_ptr(), CAP, A, probs.data_ptr(), value.data_ptr(),
                                  d_dense_ptr=d_fb.data_ptr())
                pb.act(probs.data_ptr(), A, cmd.data_ptr(), seed=3)
                g.step_device(cmd.data_ptr(), 1)
                g.done_device(d_new.data_ptr())
                pb.reset_memory(d_new.data_ptr())
                g.synchronize()
                restarted += int((d_new != 0).sum().item())
        if step % 50 == 49 or step == STEPS - 1:
            for (_, g, pb, _, _) in worlds:
                g.synchronize(), pb.synchronize()
            assert np.array_equal(worlds[0][1].digest(), worlds[1][1].digest()), "step %d" % step
            assert torch.equal(worlds[0][4][0], worlds[1][4][0]) and torch.equal(worlds[0][4][2], worlds[1][4][2])
    for b in range(0, A, 97):
        h0, a0 = worlds[0][2].get_memory(b)
        h1, a1 = worlds[1][2].get_memory(b)
        # (the separate calls have already reset the restarted agents' stored memory; predict does that when it next runs)
        if not bool(d_new[b].item()):
            assert np.array_equal(h0, h1) and np.array_equal(a0, a1)
    assert restarted > 20
