"""Driver for oracle/_ref/sf_ref_tick: the reference's own tick path compiled head-less (oracle/ref_tick.py: the
client's headers with only the SFML / keyboard / menu member functions blanked, every other line unedited).
Test infrastructure only.  available() is False where the binary was never built or the reference's data files
(Items/, character/) are not on this machine."""
import os
import shutil
import subprocess
import tempfile

import numpy as np

from strikeforce_amd import abi, config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "sf_ref_tick")
REF_CLIENT = "/root/reference/StrikeForce-client"
MODES = {abi.MODE_SOLO: "Solo", abi.MODE_TIMER: "Timer", abi.MODE_SQUAD: "Squad", abi.MODE_BATTLE: "Battle_Royal"}


NATIVE = (3, 30, 100, 9000, 9000, 9000, 9000)  # gameplay.hpp:37


def available():
    return os.path.exists(BIN) and os.path.isdir(os.path.join(REF_CLIENT, "Items"))


def binary(dims, squad_agents=False):
    """The build for these dimensions: the native one as __graft_entry__.build() left it; a patched-dimensions one
    (gameplay.hpp:37 replaced, oracle/ref_tick.py) is compiled on first use where the checkout exists."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_tick
    path = ref_tick.binary_for(dims, squad_agents)
    if path != BIN and os.path.isdir(REF_CLIENT):
        ref_tick.build(quiet=True, dims=dims, squad_agents=squad_agents)
    return path if os.path.exists(path) else None


class RefTick:
    """One reference game in a child process.  `workload`: a config.Workload with the reference's native dimensions
    (3 x 30 x 100, gameplay.hpp:37) or any other (a patched-dimensions build), mode Solo / Timer / Squad; its map is written out as map/floor1-3.txt in the
    reference's own text format and read back by gameplay::setup()."""

    def __init__(self, workload, player_tokens, agents=False, native_caps=True, squad_agents=False):
        """native_caps: the reference's own slot pools of 9000 (the run must stay within the configuration's caps:
        `over`); False: a build whose pools are the configuration's, so that they run dry at the same moment."""
        cfg = workload.cfg
        dims = (cfg.floors, cfg.rows, cfg.cols)
        if native_caps:
            dims += (9000, 9000, 9000, 9000)
        else:  # (the reference pools exits by B as well, `portal[B]` gameplay.hpp:51-53: `over` counts exits beyond cap_portals)
            dims += (cfg.cap_humans, cfg.cap_zombies, cfg.cap_bullets, cfg.cap_chests)
        self.p = None
        exe = binary(dims, squad_agents)
        assert exe, "no sf_ref_tick build for %s" % (dims,)
        self.cfg = cfg
        self.dir = tempfile.mkdtemp(prefix="sf_reftick_run_")
        os.makedirs(os.path.join(self.dir, "map"))
        for d in ("Items", "character"):
            os.symlink(os.path.join(REF_CLIENT, d), os.path.join(self.dir, d))
        cells = cfg.rows * cfg.cols
        chars = bytes(workload._map.raw).decode("ascii")
        portal = list(workload._portal)
        for f in range(cfg.floors):
            with open(os.path.join(self.dir, "map", "floor%d.txt" % (f + 1)), "w") as fh:
                fh.write(config.format_floor_text(chars[f * cells:(f + 1) * cells], portal[f * cells:(f + 1) * cells],
                                                  cfg.rows, cfg.cols))
        blob = "player\n" + "\n".join(str(int(t)) for t in player_tokens) + "\n"
        with open(os.path.join(self.dir, "profile.txt"), "w") as fh:
            fh.write(blob)
        # the account file Client::give_info sends to the match server (gameplay.hpp:120-131); user = "ref_tick"
        os.makedirs(os.path.join(self.dir, "accounts", "game", "ref_tick"))
        with open(os.path.join(self.dir, "accounts", "game", "ref_tick", "info, ref_tick.txt"), "w") as fh:
            fh.write(blob)
        self.p = subprocess.Popen([exe], cwd=self.dir, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)
        r = self._cmd("init profile.txt %s %d %d" % (MODES[cfg.mode], cfg.level, int(agents)))
        assert r and r[0] == "ok dims " + " ".join(str(d) for d in dims), r

    def _cmd(self, line):
        self.p.stdin.write(line + "\n")
        self.p.stdin.flush()
        out = []
        while True:
            s = self.p.stdout.readline()
            if not s:
                raise RuntimeError("sf_ref_tick died on %r (exit %s)" % (line, self.p.poll()))
            s = s.rstrip("\n")
            if s == "end":
                return out
            out.append(s)

    def close(self):
        if self.p:
            try:
                self.p.stdin.write("quit\n")
                self.p.stdin.flush()
                self.p.wait(timeout=10)
            except Exception:  # noqa: BLE001
                self.p.kill()
            self.p = None
            shutil.rmtree(self.dir, ignore_errors=True)

    def __del__(self):
        self.close()

    def reset(self, tb, serial):
        assert self._cmd("reset %d %d" % (tb, serial)) == ["ok"]

    def logging(self, on):
        assert self._cmd("logging %d" % int(on)) == ["ok"]

    def join_match(self, host, port, password):
        """Battle_Royal: setup() -> load_data() asks for the server's IP, port and password on stdin and joins the match
        (gameplay.hpp:1806-1830).  Blocks until the server has all players.  Returns (tb, serial, ind, n, team)."""
        self.p.stdin.write("reset_native\n%s\n%d\n%s\n" % (host, port, password))
        self.p.stdin.flush()
        out = []
        while True:
            ln = self.p.stdout.readline()
            if not ln:
                raise RuntimeError("sf_ref_tick died while joining the match")
            if ln.rstrip("\n") == "end":
                break
            out.append(ln.rstrip("\n"))
        t = [int(x) for x in out[-1].split("ok ")[-1].split()]
        assert t[5] == 0, "the reference client could not connect: %r" % out
        return tuple(t[:5])

    def reset_native(self, replay_path=None):
        """setup() with its own seeds (time(), libc rand) — or, with replay_path, in replay mode from that .sf_sample:
        load_data asks for the path on stdin (gameplay.hpp:1750-1763).  Returns (tb, serial_number) as the game has them."""
        if replay_path is not None:
            assert self._cmd("replay 1") == ["ok"]
            self.p.stdin.write("reset_native\n%s\n" % replay_path)
            self.p.stdin.flush()
            out = []
            while True:
                ln = self.p.stdout.readline()
                if not ln:
                    raise RuntimeError("sf_ref_tick died in replay setup")
                if ln.rstrip("\n") == "end":
                    break
                out.append(ln.rstrip("\n"))
            r = [x for x in out if x.startswith("ok ") or x.endswith(" ok")] or out[-1:]
            # load_data's prompt ("Enter the file's address: ") shares the line with the answer
            t = out[-1].split("ok ")[-1].split()
        else:
            r = self._cmd("reset_native")
            t = r[-1].split()[1:]
        return int(t[0]), int(t[1])

    def rivals_are_dead(self):
        return bool(int(self._cmd("rivals")[0].split()[1]))

    def logclose(self):
        r = self._cmd("logclose")
        return os.path.join(self.dir, r[0].split(" ", 1)[1])

    def step(self, chars):
        """chars: bytes/str, chars[0] the player's command, chars[k] agent k's scripted action."""
        if isinstance(chars, (bytes, bytearray, np.ndarray)):
            chars = bytes(chars).decode("ascii")
        r = self._cmd("step " + chars)
        assert len(r) == 1 and r[0].startswith("ok "), r
        # generator draws of this step by phase: zombie_action, update_bull (1st), human_action, update_bull (2nd),
        # the next loop top's spawns, everything else
        t = [int(x) for x in r[0].split()[1:]]
        self.phase_draws = t[:6]
        # what the reference's own check_end() returned at the loop top behind this step (gameplay.hpp:1450; its Timer
        # branch reads time(0), gameplay.hpp:1145: not meaningful under a fixed tb)
        self.ended = bool(t[6])

    def calls(self):
        """[(agent id, 'N'ew|'D'eleted|'P'redict|'U'pdate, a, b, frame)] since the last call."""
        out = []
        for s in self._cmd("calls"):
            t = s.split()
            out.append((int(t[1]), t[2], int(t[3]), int(t[4]), int(t[5])))
        return out

    def _obs(self, lines):
        n = int(lines[0].split()[1])
        o = np.zeros(n, dtype=np.uint32)
        for s in lines[1:]:
            i, v = s.split(":")
            o[int(i, 16)] = int(v, 16)
        return o.view(np.float32)

    def last_obs(self, agent_id):
        return self._obs(self._cmd("obs %d" % agent_id))

    def observe(self, slot):
        """gameplay::bot's encoding (Custom.hpp:137-159) for hum[slot] as of now: 30752 floats."""
        return self._obs(self._cmd("observe %d" % slot))

    def dump(self):
        """The state as arrays in the word order of include/strikeforce.h's dump records: {"hdr": int64 [frame, kills,
        teams_kills, loot, chests, jomle, steps, rng[18]], "humans" [H,28] (column 3 `profile` = -1: no counterpart),
        "zombies" [Z,7], "bullets" [B,11], "portals" [P,4], "flags" u8[cells], "dmg", "pidx" i32[cells]}.
        self.over = live entities beyond the configuration's caps (the reference's own caps are 9000)."""
        cfg = self.cfg
        H, Z, B, P = cfg.cap_humans, cfg.cap_zombies, cfg.cap_bullets, cfg.cap_portals
        cells = cfg.floors * cfg.rows * cfg.cols
        out = {"dmg": np.zeros(cells, np.int32), "pidx": np.full(cells, -1, np.int32)}
        for s in self._cmd("dump %d %d %d %d" % (H, Z, B, P)):
            tag, _, rest = s.partition(" ")
            if tag == "hdr":
                v = np.array(rest.split(), dtype=np.int64)
                self.ind = int(v[7])
                out["hdr"] = np.concatenate([v[:7], v[8:26]])
            elif tag in "HZBP":
                name, width = {"H": ("humans", 28), "Z": ("zombies", 7), "B": ("bullets", 11), "P": ("portals", 4)}[tag]
                out[name] = np.array(rest.split(), dtype=np.int32).reshape(-1, width)
            elif tag == "A":
                self.active_agents = [int(x) for x in rest.split()]
            elif tag == "over":
                self.over = int(rest)
            elif tag == "F":
                out["flags"] = np.frombuffer(bytes.fromhex(rest), dtype=np.uint8).copy()
            elif tag == "S":
                for t in rest.split():
                    i, v = t[1:].split(":")
                    out["dmg" if t[0] == "d" else "pidx"][int(i)] = int(v)
        return out


HDR_NAMES = ["frame", "kills", "teams_kills", "loot", "chests", "jomle", "steps"] + ["rng[%d]" % i for i in range(18)]


def arrays_of(d):
    """An oracle_lib.ArenaDump in the same form as RefTick.dump()."""
    h = d.hdr
    hdr = np.array([h.frame, h.kills, h.teams_kills, h.loot, h.chests, h.jomle, h.steps] + list(h.rng), dtype=np.int64)

    def table(recs, cls):
        w = abi.C.sizeof(cls) // 4
        return np.concatenate([np.frombuffer(bytes(r), dtype=np.int32) for r in recs]).reshape(-1, w)

    return {"hdr": hdr, "humans": table(d.humans, abi.HumanRec), "zombies": table(d.zombies, abi.ZombieRec),
            "bullets": table(d.bullets, abi.BulletRec), "portals": table(d.portals, abi.PortalRec),
            "flags": d.flags, "dmg": d.dmg, "pidx": d.pidx}


def first_difference(ref, ours):
    """None if the reference's dump equals ours (the `profile` column of the humans, which the reference lacks, aside),
    else a readable description of the first difference."""
    for key in ("hdr", "humans", "zombies", "bullets", "portals", "flags", "dmg", "pidx"):
        a, b = np.asarray(ref[key]), np.asarray(ours[key])
        if key == "humans":
            a = a.copy()
            a[:, 3] = b[:, 3]
        if a.shape != b.shape:
            return "%s: shape %s != %s" % (key, a.shape, b.shape)
        if not np.array_equal(a, b):
            idx = tuple(int(x[0]) for x in np.nonzero(a != b))
            names = {"hdr": HDR_NAMES, "humans": [n for n, _ in abi.HumanRec._fields_[:17]] + ["cons%d" % i for i in range(4)]
                     + ["throw_cnt%d" % i for i in range(4)] + ["blocks", "portals", "portal_ind"],
                     "zombies": [n for n, _ in abi.ZombieRec._fields_], "bullets": [n for n, _ in abi.BulletRec._fields_],
                     "portals": [n for n, _ in abi.PortalRec._fields_]}.get(key)
            what = "%s%s" % (key, list(idx)) if names is None else \
                ("%s.%s" % (key, names[idx[0]]) if a.ndim == 1 else "%s[%d].%s" % (key, idx[0], names[idx[1]]))
            return "%s: reference %s, ours %s" % (what, a[idx], b[idx])
    return None
