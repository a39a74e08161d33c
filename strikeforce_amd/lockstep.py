"""Lock-step wire adapter: play a Battle-Royal match through the reference's match server.

The reference's online mode has no authoritative simulation: the server (StrikeForce-server/server.cpp) hands every
client the same seed and its player index, forwards the account blobs, and then only relays one command char per
player per loop iteration; every client simulates the whole world (gameplay.hpp:66-193,1795-1859).  This module is the
client side of that protocol with the GPU simulator (or, in tests, the oracle / the wave emulator) as the world.

Wire format (all strings NUL-terminated, read byte-wise like basic.hpp my_recv):
  -> password                              <- "A" | "R"                      Client::start   gameplay.hpp:66-111
  <- "<tb> <serial>"      <- "<n> <ind> <team>"                             server.cpp:239-246
  -> own account blob ("name\\nHp\\n..."), <- for every other player in index order: blob, team   gameplay.hpp:120-151
  per iteration: -> own command char;   <- the command of every other player whose client was alive when the
  iteration began, in index order        send_it / recieve gameplay.hpp:113-118,170-193; server.cpp:76-117
  leaving: '~' when the own player is dead, '+' when all rivals are dead, '_' to quit   gameplay.hpp:1102-1143,939-953

One deliberate difference from the reference client: it skips players that are dead *when human_action runs*
(`if(!mh[i]) continue;` gameplay.hpp:172), but the server has already relayed the command of a player that died in
the first half of that very iteration, so the reference client falls out of step with the byte stream in that case.
This adapter reads what the server sends: the commands of the players alive at the *start* of the iteration.

Round-1 limitation: every player must use the same character record (the simulator takes one player profile).
"""
import ctypes as C
import socket

import numpy as np

from . import abi, config


class ProtocolError(RuntimeError):
    pass


def format_blob(name, tokens):
    """The account file the reference sends in give_info(): the 33 whitespace-separated tokens, one per line."""
    return "\n".join([name] + [str(int(t)) for t in tokens])


def parse_blob(text):
    tok = text.split()
    if len(tok) != 33:
        raise ProtocolError("an account blob has a name and 32 integers, got %d tokens" % len(tok))
    return tok[0], [int(x) for x in tok[1:]]


class MatchClient:
    def __init__(self, host, port, password, profile_tokens, name="mi355x", timeout=5.0):
        self.addr, self.password = (host, int(port)), password
        self.name, self.tokens = name, [int(t) for t in profile_tokens]
        self.timeout = timeout
        self.sock = None

    # -- byte-level helpers (basic.hpp:261-271 my_recv) ---------------------------------------------------------
    def _recv_cstr(self):
        out = bytearray()
        while True:
            b = self.sock.recv(1)
            if not b:
                raise ProtocolError("server closed the connection")
            if b == b"\0":
                return out.decode("ascii", "replace")
            out += b

    def _send_cstr(self, s):
        self.sock.sendall(s.encode("ascii") + b"\0")

    # -- handshake ----------------------------------------------------------------------------------------------
    def connect(self):
        self.sock = socket.create_connection(self.addr, timeout=self.timeout)
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self._send_cstr(self.password)
        if self._recv_cstr()[:1] != "A":
            raise ProtocolError("wrong password")
        self.tb, self.serial = (int(x) for x in self._recv_cstr().split())
        self.n, self.ind, self.team = (int(x) for x in self._recv_cstr().split())
        self._send_cstr(format_blob(self.name, self.tokens))
        self.teams = [0] * self.n
        self.teams[self.ind] = self.team
        self.records = [None] * self.n  # every player's character record (get_info, gameplay.hpp:120-151)
        self.records[self.ind] = list(self.tokens)
        for i in range(self.n):
            if i == self.ind:
                continue
            _, tok = parse_blob(self._recv_cstr())
            self.teams[i] = int(self._recv_cstr())
            self.records[i] = list(tok)
        return self

    def workload(self, rows, cols, map_bytes, portal=None, floors=1, H=64, Z=64, B=256, P=32, chests=9000, device=0):
        """The one-arena Battle workload of this match as seen by this client (`ind` = the server-assigned index)."""
        cfg = config.make_config(1, rows, cols, floors=floors, H=H, Z=Z, B=B, P=P, chests=chests,
                                 mode=abi.MODE_BATTLE, level=1, n_agents=self.n, teams=self.teams, auto_reset=0,
                                 player_tokens=self.tokens, device=device, ind=self.ind,
                                 agent_tokens=self.records)  # every player is built from the record it sent
        return config.Workload("match", cfg, map_bytes, portal or [-1] * (floors * rows * cols))

    # -- per iteration ------------------------------------------------------------------------------------------
    def send_command(self, ch):
        self.sock.sendall(bytes([ord(ch) if isinstance(ch, str) else int(ch), 0]))

    def recv_commands(self, expected):
        """One command for every player index in `expected` (ascending), as the server relays them."""
        out = {}
        for i in expected:
            s = self._recv_cstr()
            out[i] = ord(s[0]) if s else ord("+")
        return out

    def close(self):
        if self.sock:
            try:
                self.sock.close()
            finally:
                self.sock = None


def play(client, sim, policy, max_iterations=100000, on_iteration=None):
    """Runs the match loop of gameplay::play() for one client.  `sim` is already built on client.workload(...);
    `policy(sim, iteration) -> command char` plays this client's human.  Returns (iterations, how it ended)."""
    tb = (C.c_uint64 * 1)(client.tb)
    sr = (C.c_uint64 * 1)(client.serial)
    sim.reset(tb, sr)
    n, ind = client.n, client.ind
    quit_seen = set()
    for it in range(max_iterations):
        alive = [bool(h.alive) for h in sim.dump(0).humans[:n]]
        if sim.done()[0]:  # check_end(): all rivals dead -> '+', own player dead -> '~'   gameplay.hpp:1103-1143
            won = alive[ind]
            client.send_command("+" if won else "~")
            client.close()
            return it, "won" if won else "died"
        mine = policy(sim, it)
        client.send_command(mine)
        if mine == "_":  # quit: the server announces it to the others and drops us   gameplay.hpp:939-953
            client.close()
            return it, "quit"
        expected = [i for i in range(n) if i != ind and alive[i] and i not in quit_seen]
        got = client.recv_commands(expected)
        cmd = np.full(n, ord("+"), dtype=np.uint8)
        cmd[ind] = ord(mine)
        for i, c in got.items():
            cmd[i] = c
            if c == ord("_"):
                quit_seen.add(i)  # its client is gone; '_' kills the player in every simulation (gameplay.hpp:696-699)
        sim.step(cmd)
        if on_iteration:
            on_iteration(it, sim)
    client.send_command("_")
    client.close()
    return max_iterations, "quit"
