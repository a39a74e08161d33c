import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(params=["folded", "layered"])
def cnn(request, monkeypatch):
    """The two forms of the policy network's convolution stack (strikeforce_policy.h): composed into one matrix (default)
    or layer by layer as the reference evaluates it (SF_POLICY_LAYERED=1, read at sf_policy_create)."""
    if request.param == "layered":
        monkeypatch.setenv("SF_POLICY_LAYERED", "1")
    else:
        monkeypatch.delenv("SF_POLICY_LAYERED", raising=False)
    return request.param
