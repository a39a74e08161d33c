"""The device source (strikeforce_amd/csrc/sf_core.hpp, sf_obs.hpp, sf_host.hpp) on the wave emulator under
AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU sanitizers are not available on the pool): every
LDS / table index the kernels form is a plain array index here, so an out-of-range cell, bitmap word or table slot that
would fault a GPU shows up as a sanitizer report.  Runs in a child process (the sanitizer runtimes must be preloaded)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
from strikeforce_amd import config
from emu_lib import Emu
for name, arenas, steps in (("C3", 3, 250), ("STRESS", 4, 400), ("MAXCAP", 2, 80), ("KITS", 2, 150), ("FLOORS", 2, 150),
                            ("C5", 2, 40), ("NATIVE", 2, 150)):
    w = config.baseline_workload(name, arenas=arenas)
    e = Emu(w, asan=True)
    e.reset(*w.seeds())
    cmds, _ = config.bench_commands(arenas, w.cfg.n_agents, steps)
    e.step_many(cmds)
    e.observe()
    e.digest()
# large pools (zombie / exit tables in the emulated LDS, sf_core.hpp ZL): every kernel variant, far enough for three words of
# zombies and two of exits, with a restart on the way (tests/test_large_pools.py's worlds)
import test_large_pools
for n in (64, 120, 160):
    w = test_large_pools.world(n, 2)
    e = Emu(w, asan=True)
    e.reset(*w.seeds())
    cmds, _ = config.bench_commands(2, 1, 4700, seed0=4321)
    e.step_many(cmds)
    e.observe()
    e.digest()
print("SANITIZED-OK")
"""


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_device_source_is_clean_under_asan_and_ubsan(tmp_path):
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no sanitizer runtimes with this gcc")
    script = tmp_path / "child.py"
    script.write_text(CHILD.format(root=ROOT))
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0 and "SANITIZED-OK" in p.stdout, (p.stdout[-1500:], p.stderr[-3000:])
