#!/usr/bin/env python3
"""ref_tick.py — TEST INFRASTRUCTURE: build oracle/_ref/sf_ref_tick, the reference's own tick path, head-less.

    python oracle/ref_tick.py [--ref /root/reference] [--quiet] [--manifest]

What cannot be had in this image is SFML, i.e. GraphicPrinter.hpp and its global `printer` (basic.hpp:41).  Everything
else on the tick path is plain C++ that compiles as it stands: the whole of random.hpp, Item.hpp, Character.hpp and
gameplay.hpp except the member functions that draw, read the keyboard or run the menus.  This recipe therefore copies
the six files below into a temporary directory with the line ranges listed in OMIT **blanked** (replaced by empty lines,
so that line numbers in the copy are the reference's own), and compiles oracle/ref_tick_main.cpp (ours) against the
copies.  Every kept line is the reference's, unedited; no stand-in is written for any omitted header, function or
object — what is omitted is simply not there, and nothing kept calls it (the compiler checks that).  Every omitted range
is a whole declaration or member function — or, inside check_end only, a whole statement that draws the end screen or
waits for the space key — and carries its reason below; anchor texts on the first and last line of each range make a
reference whose lines have moved fail loudly.  The temporary directory is deleted; the binary lands
in oracle/_ref/ (git-ignored).  Without a checkout nothing is built (exit code 0).

The one class the reference leaves to the user is `Agent` (random.hpp:25 -> selected_agent.hpp:25 picks a bot's
Agent.hpp; bots/bot-0/Agent.hpp:27-37 is the minimal form): ref_tick_main.cpp supplies a scripted one, which is the
plug-in contract, not a stand-in.  What ref_tick_main.cpp restates from the omitted play() is the ORDER of its calls
(gameplay.hpp:1441-1471, fifteen lines) and get_my_action's two lines that fetch the player's command
(gameplay.hpp:955-960, incl. client.send_it() in an online match).  check_end (gameplay.hpp:1102-1229) is kept: its
comparisons, its bookkeeping and its return value; of its lines only `render_it();`, `if(printThread.joinable())
printThread.join();`, `printer.print(...)` and `while(getch() != ' ');` are blanked (eleven groups, OMIT below), so SURVEY
§8 row a19 is pinned on the reference's own function (tests/test_ref_check_end.py; its Timer branch reads time(0) and is
not comparable under a fixed tb).

Pins (tests/test_ref_tick.py): rows a4-a20 — Backpack/Human, gen_human, node/showit, the slot allocators, the
three spawns, zombie_action, portal_damage, update_tmp, hit_human/hit_zombie, update_bull, human_action / get_command /
human_rnpc_bot, obey, teleport, claim_chest, setup/load_data (Solo, Timer, Squad), check_end, describe + gameplay::bot — on the
reference's native world (gameplay.hpp:37: 3 floors x 30 x 100, its caps of 9000 never reached).  Also: load_data's
logging / replay branches (.sf_sample, tests/test_ref_replay.py) and its online branch + class Client, in a real match
through the reference's own server (tests/test_lockstep_server.py).

Patched dimensions (second flavour, `build(dims=(F, N, M, H, Z, B, C))` -> oracle/_ref/sf_ref_tick_<dims>): the
reference's world size and slot-pool caps are compile-time constants on ONE line, gameplay.hpp:37.  For BASELINE.json's
configurations (32x32 ... 128x128, caps of 1-64) that line — and only that line — is blanked as well and
ref_tick_main.cpp declares the same nine constants with the requested values (`lim_portal` / `lim_block` keep the
reference's 1000 / 1100): the procedure SURVEY §8c prescribes ("G:37 is the only line to change").  With the caps equal
to the configuration's, the slot allocators running dry is pinned too.  The native build stays the primary pin.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CLIENT = "StrikeForce-client"
OUT = os.path.join(HERE, "_ref", "sf_ref_tick")

# file -> [(first, last, text on the first line, text on the last line, reason)]; lines are 1-based, inclusive
OMIT = {
    "basic.hpp": [
        (41, 41, '#include "GraphicPrinter.hpp"', '#include "GraphicPrinter.hpp"', "SFML (<SFML/Graphics.hpp>)"),
        (47, 50, "void usleep(int x){", "}", "sf::sleep"),
        (241, 259, "std::string head(bool ingame = false, bool dont = false){", "}", "printer.cls()"),
    ],
    "macros.hpp": [],
    "random.hpp": [
        (25, 25, '#include "selected_agent.hpp"', '#include "selected_agent.hpp"',
         "the Agent plug-in slot (bot-0.5's Agent needs curl + a key press at construction, Agent.hpp:66-71); "
         "ref_tick_main.cpp supplies the Agent class"),
    ],
    "Item.hpp": [(25, 25, '#include "random.hpp"', '#include "random.hpp"', "local include: the copies are included in order by ref_tick_main.cpp")],
    "Character.hpp": [
        (25, 25, '#include "Item.hpp"', '#include "Item.hpp"', "local include"),
        (168, 222, "void show(int money, bool b = false, bool ingame = false) const{", "}", "Backpack::show: head(), printer, getch"),
        (346, 349, "void show_backpack(bool b = false, bool ingame = false) const{", "}", "calls Backpack::show"),
    ],
    "gameplay.hpp": [
        (25, 25, '#include "Character.hpp"', '#include "Character.hpp"', "local include"),
        (823, 927, "void my_command(){", "}", "keyboard (kbhit/getch) and command_list()"),
        (939, 963, "void get_my_action(){", "}", "my_command(), render_it(), printer"),
        (1014, 1057, "void command_list(bool b = false){", "}", "printer"),
        # check_end (gameplay.hpp:1102-1229) stays — its comparisons, its bookkeeping (money, level-ups, the online
        # player's last send) and its return value are what play() acts on (`if(check_end()) break;` :1450) — without
        # the statements that draw the end screen and wait for the space key: whole statements on their own lines,
        # none of them the sole body of an `if`
        (1114, 1118, "render_it();", "while(getch() != ' ');", "check_end, match won: end screen + key wait"),
        (1123, 1126, "render_it();", 'printer.print("You\'re disconnected :(\\npress space button to continue\\n");', "check_end, disconnected: end screen"),
        (1128, 1128, "while(getch() != ' ');", "while(getch() != ' ');", "check_end, disconnected: key wait"),
        (1138, 1142, "render_it();", "while(getch() != ' ');", "check_end, player dead: end screen + key wait"),
        (1149, 1153, "render_it();", "while(getch() != ' ');", "check_end, Timer lost: end screen + key wait"),
        (1167, 1170, "render_it();", "printer.print(s);", "check_end, Timer won: end screen"),
        (1176, 1176, "while(getch() != ' ');", "while(getch() != ' ');", "check_end, Timer won: key wait"),
        (1193, 1196, "render_it();", "printer.print(s);", "check_end, Solo won: end screen"),
        (1202, 1202, "while(getch() != ' ');", "while(getch() != ' ');", "check_end, Solo won: key wait"),
        (1216, 1219, "render_it();", "printer.print(s);", "check_end, Squad won: end screen"),
        (1225, 1225, "while(getch() != ' ');", "while(getch() != ' ');", "check_end, Squad won: key wait"),
        (1420, 1426, "void render_it(){", "}", "starts print_game on a thread"),
        (1428, 1505, "void play(){", "}", "printer.start/stop, get_my_action, check_end, render_it; its call ORDER is restated in ref_tick_main.cpp"),
        (1507, 1678, "void open(){", "}", "menus: head(), usleep, play()"),
        (1944, 2067, "void gameplay::print_game() const{", "}", "printer.render, head(), usleep"),
    ],
    "bots/bot-0.5/Custom.hpp": [(25, 25, '#include "../../gameplay.hpp"', '#include "../../gameplay.hpp"', "local include")],
}
# lines that must be where the recipe thinks they are (the constructs the pins rest on)
CONTEXT = [
    ("gameplay.hpp", 37, "int constexpr F = 3, N = 30, M = 100, H = 9000, Z = 9000, B = 9000, C = 9000, lim_portal = 1000, lim_block = 1100;"),
    ("gameplay.hpp", 437, "struct gameplay{"),
    ("gameplay.hpp", 654, "void zombie_action(){"),
    ("gameplay.hpp", 695, "void obey(const char c, Environment::Character::Human &player){"),
    ("gameplay.hpp", 965, "void human_action(){"),
    ("gameplay.hpp", 1059, "void update_bull(){"),
    ("gameplay.hpp", 1102, "bool check_end(){"),
    ("gameplay.hpp", 1229, "}"),
    ("gameplay.hpp", 1231, "void setup(){"),
    ("gameplay.hpp", 1739, "} g;"),
    ("gameplay.hpp", 1741, "void gameplay::load_data(){"),
    ("gameplay.hpp", 1927, "char gameplay::human_rnpc_bot(Environment::Character::Human& player) const{"),
    ("Character.hpp", 289, "class Human: public Character{"),
    ("Character.hpp", 873, "void gen_human(bool rnpc, Human &h, int lvl, std::vector<int> cor_, std::string name = \"\", std::string dir = \"\"){"),
    ("bots/bot-0.5/Custom.hpp", 29, "std::vector<float> describe(const node &cell, const Environment::Character::Human &player){"),
    ("bots/bot-0.5/Custom.hpp", 137, "char gameplay::bot(Environment::Character::Human& player) const {"),
]


def copy_name(rel):
    return "ref_" + rel.replace("/", "_").replace("-", "_").replace(".hpp", ".inc")


def manifest():
    rows = []
    for rel, om in OMIT.items():
        for first, last, a0, _a1, why in om:
            rows.append("%-24s %4d-%-4d %-60s %s" % (rel, first, last, a0[:60], why))
    return "\n".join(rows)


DIMS_LINE = ("gameplay.hpp", 37)


NATIVE_DIMS = (3, 30, 100, 9000, 9000, 9000, 9000)
BENCH_DIMS = (1, 64, 64, 8, 24, 64, 9000)  # BASELINE.json configs[2]: 64x64, H8 Z24 B64 (bench.py cpu_baseline, kind "reference")
# the other dimension sets the tests play (configs[0], configs[1], STRESS, the Squad and three-floor cases, the one-slot
# bullet pool of tests/test_order_scenarios.py): the
# default call compiles them too, side by side, so that the first test run does not compile them one after the other
TEST_DIMS = [(1, 32, 32, 1, 4, 16, 9000), (1, 64, 64, 1, 16, 32, 9000), (1, 24, 40, 6, 12, 5, 6),
             (1, 28, 36, 12, 10, 48, 9000), (3, 20, 30, 12, 10, 32, 9000), (3, 30, 100, 12, 4, 1, 4),
             # configs[3] / configs[4]'s dimensions: the maps whose flag plane the device keeps in HBM (k_step<., true, .>)
             (1, 128, 128, 10, 20, 64, 9000), (1, 256, 256, 8, 56, 128, 9000)]


def binary_for(dims=None, squad_agents=False):
    out = OUT + ("_squadagents" if squad_agents else "")
    if dims is None or tuple(dims) == NATIVE_DIMS:
        return out
    return out + "_" + "_".join(str(int(d)) for d in dims)


def build(ref="/root/reference", quiet=False, dims=None, squad_agents=False):
    """dims None: the reference as it stands.  dims (F, N, M, H, Z, B, C): gameplay.hpp:37 blanked too, its constants
    declared with these values by ref_tick_main.cpp (-DSF_REF_DIMS).  squad_agents: compiled with the reference's own
    switch USE_AGENT_IN_SQUAD_NPCS (gameplay.hpp:1883,1896: every Squad human gets an Agent)."""
    base = os.path.join(ref, CLIENT)
    out = binary_for(dims, squad_agents)
    patched = not (dims is None or tuple(dims) == NATIVE_DIMS)
    if not os.path.isfile(os.path.join(base, "gameplay.hpp")):
        if not quiet:
            print("no reference checkout at %s: %s left as is" % (ref, os.path.relpath(out, os.path.dirname(HERE))))
        return None
    main = os.path.join(HERE, "ref_tick_main.cpp")
    srcs = [os.path.join(base, rel) for rel in OMIT] + [main, os.path.abspath(__file__)]
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(p) for p in srcs):
        return out
    tmp = tempfile.mkdtemp(prefix="sf_reftick_")
    try:
        kept_total = omitted_total = 0
        for rel, om in OMIT.items():
            with open(os.path.join(base, rel), encoding="utf-8", errors="replace", newline="") as f:
                lines = f.read().replace("\r\n", "\n").split("\n")

            def check(no, text):
                got = lines[no - 1].strip()
                if got != text:
                    raise SystemExit("ref_tick: %s:%d is %r, expected %r — the reference's lines have moved" % (rel, no, got, text))

            for r2, no, text in CONTEXT:
                if r2 == rel:
                    check(no, text)
            om = list(om)
            if patched and rel == DIMS_LINE[0]:
                text = [t for r2, no, t in CONTEXT if r2 == rel and no == DIMS_LINE[1]][0]
                om.append((DIMS_LINE[1], DIMS_LINE[1], text, text, "patched dimensions"))
            for first, last, a0, a1, _why in om:
                check(first, a0)
                check(last, a1)
                for i in range(first - 1, last):
                    lines[i] = ""
                omitted_total += last - first + 1
            kept_total += len(lines)
            # CRLF -> LF is the only byte-level change (a backslash-CR at a line end would otherwise not continue a line)
            with open(os.path.join(tmp, copy_name(rel)), "w", encoding="utf-8", newline="") as f:
                f.write("\n".join(lines) + "\n")
        # basic.hpp:43 `#include "macros.hpp"` resolves beside the copy
        shutil.copy(os.path.join(tmp, copy_name("macros.hpp")), os.path.join(tmp, "macros.hpp"))
        os.makedirs(os.path.dirname(out), exist_ok=True)
        cmd = ["g++", "-std=c++17", "-O2", "-w", "-pthread", "-I", tmp, main, "-o", out]
        if patched:
            cmd += ["-DSF_REF_DIMS=%s" % ",".join(str(int(d)) for d in dims)]
        if squad_agents:
            cmd += ["-DUSE_AGENT_IN_SQUAD_NPCS"]
        subprocess.check_call(cmd)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if not quiet:
        print("built oracle/_ref/%s from %s (%d lines of 7 client files, %d of them blanked: SFML / keyboard / menus%s)"
              % (os.path.basename(out), ref, kept_total, omitted_total, " + gameplay.hpp:37, the dimensions" if patched else ""))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default=os.environ.get("REF", "/root/reference"))
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--manifest", action="store_true", help="print the omitted ranges and why")
    ap.add_argument("--dims", default="", help="F,N,M,H,Z,B,C: also blank gameplay.hpp:37 and compile for these")
    ap.add_argument("--squad-agents", action="store_true", help="compile with -DUSE_AGENT_IN_SQUAD_NPCS")
    a = ap.parse_args()
    if a.manifest:
        print(manifest())
    else:
        build(a.ref, a.quiet, [int(x) for x in a.dims.split(",")] if a.dims else None, a.squad_agents)
        if not a.dims and not a.squad_agents:  # the default call (oracle/Makefile `ref`) also makes the Squad-agents flavour,
            # the one bench.py times as its cpu_baseline (BASELINE configs[2]) and the tests' dimension sets
            from concurrent.futures import ThreadPoolExecutor
            jobs = [(None, True), (BENCH_DIMS, False)] + [(d, False) for d in TEST_DIMS]
            with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
                list(ex.map(lambda j: build(a.ref, a.quiet, j[0], j[1]), jobs))
    sys.exit(0)
