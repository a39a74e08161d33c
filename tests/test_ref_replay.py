"""`.sf_sample` against the reference's own writer and reader, compiled in oracle/_ref/sf_ref_tick
(gameplay.hpp:1749-1794 load_data's logging / replay branches, :966-969 human_action, Character.hpp:570-648
Human::scan_file / log_file): a game LOGGED by the reference is read by strikeforce_amd.replay and replayed on the oracle
to the reference's final state; a sample WRITTEN by strikeforce_amd.replay is read and replayed by the reference to the
oracle's final state.  Skipped where the reference build is not there."""
import ctypes as C
import os

import numpy as np
import pytest

import ref_cases
import reftick
from emu_lib import Emu
from oracle_lib import Oracle
from strikeforce_amd import abi, config, replay

pytestmark = pytest.mark.skipif(not reftick.available(), reason="oracle/_ref/sf_ref_tick not built (no reference checkout)")
RICH = ref_cases.RICH
STRONG1 = list(RICH[:3]) + [1, 1, 1] + list(RICH[6:])  # the same record at level 1 in every mode: Human::build applies no level-ups


def _commands(n, seed):
    rng = np.random.RandomState(seed)
    return "".join(abi.BENCH_COMMANDS[i] for i in rng.randint(0, 28, size=n))


def _logged_game(w, player, cmds):
    r = reftick.RefTick(w, player)
    r.logging(True)
    tb, serial = r.reset_native()  # the reference's own seeds: time(0) and three libc rand() calls
    for ch in cmds:
        r.step(ch)
    final = r.dump()
    assert r.over == 0
    path = r.logclose()
    text = open(path).read()
    r.close()
    return tb, serial, final, path, text


@pytest.mark.parametrize("mode,player", [(abi.MODE_SOLO, STRONG1), (abi.MODE_TIMER, RICH)],
                         ids=["solo-level1-record", "timer-level10-account"])
def test_a_game_logged_by_the_reference_replays_here(mode, player, tmp_path):
    """The reference logs a 400-step game; the reference itself replays the file (replay_mode) and this repo replays it
    (strikeforce_amd.replay on oracle and emulator): same final state, whole dump.
    A quirk the two replays share: Human::log_file writes def_Hp / mindamage_def / def_stamina AFTER the level-ups that
    Human::build applied (Character.hpp:619-648,691-707), and scan_file applies them again on reading
    (Character.hpp:570-617) — so the replay of a levelled account starts with Hp 16350 / mindamage 1135 where the
    logged game had 15000 / 1000.  For a level-1 record the replay IS the logged game (asserted)."""
    w = ref_cases.native(mode, 3, player, maps="shipped")
    cmds = _commands(400, 5)
    tb, serial, final_logged, path, text = _logged_game(w, player, cmds)
    assert path.endswith(".sf_sample") and "online:0-lvl:3" in path  # gameplay.hpp:1785-1793
    sample_copy = tmp_path / "logged.sf_sample"
    sample_copy.write_text(text)
    s = replay.read_sample(str(sample_copy))
    assert (s.tb, s.serial, s.ind, s.team, s.commands) == (tb, serial, 0, 1, cmds)
    levelled = list(player) == list(RICH)
    if levelled:  # 27 level-ups of +50 / +5 / +50 (levels 10, 10, 10: Character.hpp:765-800)
        assert s.profile_tokens[:3] == [15000 + 27 * 50, 1000 + 27 * 5, 15000 + 27 * 50] and s.profile_tokens[3:] == list(player)[3:]
    else:
        assert s.profile_tokens == list(player)
    # the reference replays its own file
    r2 = reftick.RefTick(w, config.HUMAN_TOKENS)  # (the record comes from the file)
    os.symlink(str(sample_copy), os.path.join(r2.dir, "logged.sf_sample"))
    tb2, serial2 = r2.reset_native(replay_path="logged.sf_sample")
    assert serial2 == serial
    for _ in cmds:
        r2.step("+")
    final_replayed = r2.dump()
    r2.close()
    if not levelled:
        assert reftick.first_difference(final_logged, final_replayed) is None
    for impl in (Oracle, Emu):
        w2 = ref_cases.native(mode, 3, s.profile_tokens, maps="shipped")
        sim = impl(w2)
        assert replay.replay(s, sim) == len(cmds)
        d = reftick.first_difference(final_replayed, reftick.arrays_of(sim.dump(0)))
        assert d is None, d


def test_a_sample_written_here_is_replayed_by_the_reference(tmp_path):
    cmds = _commands(350, 9)
    s = replay.Sample(1771155561, 1073741823, RICH, cmds, name="player")
    w = ref_cases.native(abi.MODE_SOLO, 2, RICH, maps="shipped")
    o = Oracle(w)
    assert replay.replay(s, o) == len(cmds)
    r = reftick.RefTick(w, config.HUMAN_TOKENS)  # the record comes from the sample, not from this one
    path = os.path.join(r.dir, "ours.sf_sample")
    replay.write_sample(path, s)
    tb, serial = r.reset_native(replay_path="ours.sf_sample")
    assert serial == s.serial
    for _ in cmds:
        r.step("+")  # replay mode: human_action takes the command from the file (gameplay.hpp:968-969)
    d = reftick.first_difference(r.dump(), reftick.arrays_of(o.dump(0)))
    assert d is None, d
    r.close()
