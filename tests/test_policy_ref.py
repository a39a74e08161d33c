"""CPU checks of the policy-network side: the batched fp32 reference against the per-agent module
built like the reference's libtorch model, the ABI surface of include/strikeforce_policy.h, and the no-GPU behaviour."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest
import torch

from strikeforce_amd import build, env, policy

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import policy_ref  # noqa: E402


def _obs(rng, B):
    """Observation-like input: mostly zeros, values of the encoder's magnitude (|x|/10)^0.2 with signs."""
    x = rng.uniform(0.0, 2.0, size=(B, 32, 31, 31)).astype(np.float32)
    x *= rng.uniform(size=x.shape) < 0.3
    x *= np.where(rng.uniform(size=x.shape) < 0.2, -1.0, 1.0).astype(np.float32)
    return x


def test_parameter_names_match_the_module():
    m = policy_ref.AgentModel()
    sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert sd == {k: tuple(v) for k, v in policy.parameter_shapes().items()}
    # 48.6 MFLOP per agent forward in the four convolutions alone
    n = sum(int(np.prod(s)) for s in sd.values())
    assert n == 160 * 32 * 9 + 3 * 160 * 160 * 9 + 2 * (2 * 480 * 160 + 2 * 480) + 160 * 329 + 160 + 2 * 3 * (160 * 160 + 160) + 160 + 1 + 9 * 160 + 9


def test_batched_reference_equals_per_agent_module():
    rng = np.random.default_rng(5)
    params = policy.init_parameters(seed=11)
    B, T = 3, 4
    models = [policy_ref.model_from_parameters(params) for _ in range(B)]
    h = np.zeros((2, B, 160), dtype=np.float32)
    a = np.zeros((B, 9), dtype=np.float32)
    a[:, 0] = 1
    with torch.no_grad():
        for t in range(T):
            obs = _obs(rng, B)
            probs, value, h = policy_ref.forward_batched(params, obs, h, a)
            for b in range(B):
                p, v = models[b](torch.from_numpy(obs[b:b + 1]))
                np.testing.assert_allclose(p.numpy(), probs[b], rtol=2e-5, atol=1e-7)
                np.testing.assert_allclose(v.numpy()[0], value[b], rtol=2e-5, atol=1e-7)
                np.testing.assert_allclose(models[b].backbone.h_state[0].view(-1).numpy(), h[0, b], rtol=2e-5, atol=1e-6)
                np.testing.assert_allclose(models[b].backbone.h_state[1].view(-1).numpy(), h[1, b], rtol=2e-5, atol=1e-6)
                act = int(rng.integers(0, 9))
                one = torch.zeros(9)
                one[act] += 1
                models[b].update_actions(one)
                a[b] = one.numpy()
    assert np.all(np.abs(probs.sum(axis=1) - 1) < 1e-5)


def test_action_weights_follow_predict():
    p = np.array([0.2, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1], dtype=np.float32)
    v = policy_ref.action_weights(p)
    assert v[0] == np.float32(0.5)
    np.testing.assert_allclose(v[1:].sum(), 0.5, rtol=1e-4)  # the eight moves share the other half


def test_policy_library_exports_header_symbols():
    build.build(verbose=False)
    L = env.load_library()
    header = open(os.path.join(ROOT, "include", "strikeforce_policy.h")).read()
    declared = set(re.findall(r"\b(sf_policy_[a-z_]+)\s*\(", header))
    assert declared == set(policy.EXPORTS), declared ^ set(policy.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    policy._bind(L)
    assert L.sf_policy_abi_version() == policy.POLICY_ABI_VERSION
    assert C.sizeof(policy.Weights) == 8 + 8 * (4 + 8 + 2 + 8 + 8)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_policy_refuses_to_run_without_a_gpu():
    with pytest.raises(env.StrikeForceError, match="no HIP device|hip"):
        policy.PolicyBatch(policy.init_parameters(0), 4)


def test_policy_rejects_bad_parameters():
    params = policy.init_parameters(0)
    bad = dict(params)
    bad["value.1.weight"] = np.zeros((2, 160), dtype=np.float32)
    with pytest.raises(ValueError, match="shape"):
        policy.PolicyBatch(bad, 4)
    del bad["value.1.weight"]
    with pytest.raises(ValueError, match="missing"):
        policy.PolicyBatch(bad, 4)


def test_restatement_reproduces_the_committed_vectors():
    """tests/golden/policy_vectors.json holds the REFERENCE network's outputs (oracle/_ref/libsf_refmodules.so, see its
    generator); the batched restatement reproduces them up to the f32 round-off of its batched reductions."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_policy_vectors
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "policy_vectors.json")))
    got = make_policy_vectors.run_restatement()
    assert [s["action_fed_back"] for s in got["steps"]] == [s["action_fed_back"] for s in want["steps"]]
    assert [s["obs_nonzero"] for s in got["steps"]] == [s["obs_nonzero"] for s in want["steps"]]
    for g, w in zip(got["steps"], want["steps"]):
        np.testing.assert_allclose(g["probs"], w["probs"], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(g["value"], w["value"], rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(g["h_abs_sum"], w["h_abs_sum"], rtol=2e-5)


def test_load_checkpoint_reads_a_torchscript_module_archive_and_a_state_dict(tmp_path):
    """policy.load_checkpoint on the two file forms a trained bot comes in: the TorchScript module archive that
    `torch::save(model, ".../model.pt")` writes (Agent.hpp:124; produced here by scripting the restated AgentModel, the
    same container format) and a Python state_dict.  Names, shapes and values must come back exactly."""
    import torch
    from strikeforce_amd import policy
    params = policy.init_parameters(seed=5)
    model = policy_ref.model_from_parameters(params)

    class Node(torch.nn.Module):  # a scriptable module tree with the same parameter names (what the C++ side registers)
        def forward(self, x: torch.Tensor) -> torch.Tensor:
            return x

    root = Node()
    for name, value in params.items():
        node, parts = root, name.split(".")
        for part in parts[:-1]:
            if not hasattr(node, part):
                node.add_module(part, Node())
            node = getattr(node, part)
        node.register_parameter(parts[-1], torch.nn.Parameter(torch.from_numpy(value.copy())))
    ts = tmp_path / "model.pt"
    torch.jit.save(torch.jit.script(root), str(ts))
    got = policy.load_checkpoint(str(ts))
    assert set(got) == set(params)
    assert all(np.array_equal(got[k], params[k]) for k in params)
    sd = tmp_path / "state.pt"
    torch.save(model.state_dict(), str(sd))
    got = policy.load_checkpoint(str(sd))
    assert all(np.array_equal(got[k], params[k]) for k in params)
    bad = dict(model.state_dict())
    bad.pop("value.1.bias")
    torch.save(bad, str(tmp_path / "bad.pt"))
    with pytest.raises(ValueError):
        policy.load_checkpoint(str(tmp_path / "bad.pt"))


def test_load_checkpoint_reads_what_libtorch_torch_save_writes(tmp_path):
    """The real C++ writer: tests/tools/make_torch_archive.cpp registers AgentModel's parameter names on a module tree
    and calls libtorch's `torch::save(module, path)` (Agent.hpp:124,159-161); policy.load_checkpoint must return every
    tensor bit-for-bit.  Needs the libtorch headers and libraries that ship inside the torch wheel."""
    import subprocess
    import torch
    from strikeforce_amd import policy
    tdir = os.path.dirname(torch.__file__)
    inc = [os.path.join(tdir, "include"), os.path.join(tdir, "include", "torch", "csrc", "api", "include")]
    if not os.path.exists(os.path.join(inc[1], "torch", "torch.h")):
        pytest.skip("no libtorch C++ headers in this torch build")
    src = os.path.join(ROOT, "tests", "tools", "make_torch_archive.cpp")
    bdir = os.path.join(ROOT, "tests", "tools", "_build")
    os.makedirs(bdir, exist_ok=True)
    exe = os.path.join(bdir, "make_torch_archive")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        cmd = ["g++", "-std=c++17", "-O1", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch.compiled_with_cxx11_abi()), src,
               "-I" + inc[0], "-I" + inc[1], "-L" + os.path.join(tdir, "lib"), "-ltorch", "-ltorch_cpu", "-lc10",
               "-Wl,-rpath," + os.path.join(tdir, "lib"), "-o", exe]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            pytest.skip("libtorch test tool does not build here: " + r.stderr[-400:])
    shapes = policy.parameter_shapes()
    names = list(shapes)
    spec = "".join("%s %s\n" % (n, " ".join(str(d) for d in shapes[n])) for n in names)
    path = str(tmp_path / "model.pt")
    r = subprocess.run([exe, path], input=spec, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1000:]
    got = policy.load_checkpoint(path)
    assert set(got) == set(names)
    for k, n in enumerate(names):
        size = int(np.prod(shapes[n]))
        i = np.arange(size, dtype=np.uint64)
        u = ((i * np.uint64(2654435761) + np.uint64(k * 40503)) & np.uint64(0xFFFFFFFF)) >> np.uint64(16)
        want = (u.astype(np.float64) / 65536.0 - 0.5).astype(np.float32).reshape(shapes[n])
        assert np.array_equal(got[n], want), n
