// ref_tick_main.cpp — TEST INFRASTRUCTURE (ours): a head-less driver around the reference's own tick path.
//
// The ref_*.inc files are the reference's basic.hpp, random.hpp, Item.hpp, Character.hpp, gameplay.hpp and
// bots/bot-0.5/Custom.hpp with the ranges listed in oracle/ref_tick.py blanked (SFML, keyboard, menus) and every other
// line unedited.  Nothing in this file restates game logic.  What it does restate, and only this:
//   * the ORDER of calls in gameplay::play()           gameplay.hpp:1441-1471   (play() itself drives the screen)
//   * get_my_action's fetch of the player's command     gameplay.hpp:955-958
//   * main()'s two start-up calls                       main.cpp:29-30
// `class Agent` is the class the reference leaves to the user (random.hpp:25 -> selected_agent.hpp:25; minimal form
// bots/bot-0/Agent.hpp:27-37; predict/update signatures bots/bot-0.5/Agent.hpp:178,217): here it plays back scripted
// actions and records what it was called with.
//
// Protocol: one command per line on stdin, one answer block per command on stdout, each block ended by a line "end".
//   init <profile file> <Solo|Timer|Squad|Battle_Royal> <level> <agents 0|1>
//                                   Battle_Royal = the online mode "Battle Royal" (gameplay.hpp:1235): reset_native then joins a
//                                   match server — load_data asks for IP, port and password on the next three stdin lines
//                                   (gameplay.hpp:1806-1817) — as the reference client does; `step` sends the player's command
//                                   (client.send_it, gameplay.hpp:959-960) and human_action receives the others' (:977-978)
//   reset <tb> <serial>             setup(); _srand(tb, serial); ++frame; loop-top spawns and end check
//   step <chars>                    one iteration of play()'s loop + the next loop top; chars[0] = the player's command,
//                                   chars[k] = the scripted action of human k's agent (where it has one); answers the six
//                                   draw counts and what check_end() returned at that loop top (`if(check_end()) break;`
//                                   gameplay.hpp:1450 — its screens and key waits are blanked, oracle/ref_tick.py)
//   dump <H> <Z> <B> <P>            state of the first H/Z/B/P slots and of every cell, in sf_*_rec word order
//   calls                           predict/update calls since the last `calls`
//   obs <agent id>                  the observation that agent's last predict() received (30752 hex words)
//   observe <slot>                  Custom.hpp's window encoding for hum[slot] now (through a probe agent)
//   logging <0|1>                   enable_logging (gameplay.hpp:438): the reference's own .sf_sample writer
//   replay <0|1>                    replay_mode: the next reset_native reads the sample's path from the NEXT stdin line
//                                   (load_data's own prompt, gameplay.hpp:1750-1763) and takes seeds, record and commands from it
//   reset_native                    setup() with the seeds it makes itself (time(), libc rand: gameplay.hpp:1233,1745-1747)
//                                   or, in replay mode, reads from the sample; answers "ok <tb> <serial>"
//   logclose                        closes the log; answers its path (relative to the working directory)
//   rivals                          gameplay::rivals_are_dead() (gameplay.hpp:497-505), check_end's helper
//   bench <steps> <seed>            timing (bench.py's cpu_baseline): `steps` iterations under the 28-command random agent
//                                   of SURVEY §8d (LCG x <- 1664525 x + 1013904223, command (x >> 16) % 28); when the
//                                   player is dead the game is set up again with tb + 1
//   quit
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

class Agent {
public:
    int id;
    Agent();
    ~Agent();
    int predict(const std::vector<float> &obs);
    void update(int action, bool imitate);
    bool in_training() { return false; }
    bool is_manual() { return false; }
};

#include "ref_basic.inc"
#ifdef SF_REF_DIMS
// Patched-dimensions flavour only (oracle/ref_tick.py): gameplay.hpp:37 is blanked in the copy and its nine constants
// are declared here with the requested sizes; lim_portal / lim_block keep the reference's values.
namespace Environment::Field {
constexpr int sf_ref_dims[7] = {SF_REF_DIMS};
int constexpr F = sf_ref_dims[0], N = sf_ref_dims[1], M = sf_ref_dims[2], H = sf_ref_dims[3], Z = sf_ref_dims[4],
              B = sf_ref_dims[5], C = sf_ref_dims[6], lim_portal = 1000, lim_block = 1100;
}
#endif
#include "ref_random.inc"
#include "ref_Item.inc"
#include "ref_Character.inc"
#include "ref_gameplay.inc"
#include "ref_bots_bot_0.5_Custom.inc"

namespace {
using namespace Environment::Field;
namespace CH = Environment::Character;

struct Call { int id; char kind; int a, b; long long frame; };
std::vector<Call> calls;
std::map<int, std::vector<float>> last_obs;
std::map<int, char> script;        // agent id -> the command char it will answer with
int next_id = 0, live_agents = 0;
bool probing = false;
std::vector<float> probe_obs;
long long steps = 0, bench_tb = 0, bench_serial = 0;
}  // namespace

Agent::Agent() : id(probing ? -1 : next_id++) {
    if (id >= 0) ++live_agents, calls.push_back({id, 'N', 0, 0, g.frame});
}
Agent::~Agent() {
    if (id >= 0) --live_agents, calls.push_back({id, 'D', 0, 0, g.frame});
}

int Agent::predict(const std::vector<float> &obs) {
    if (probing) { probe_obs = obs; return 0; }
    last_obs[id] = obs;
    char c = script.count(id) ? script[id] : '+';
    int act = 0;
    for (int i = 0; i < (int)g.action.size(); ++i)
        if (g.action[i] == c) act = i;
    calls.push_back({id, 'P', act, (int)obs.size(), g.frame});
    return act;
}

void Agent::update(int action, bool imitate) { calls.push_back({id, 'U', action, (int)imitate, g.frame}); }

namespace {

void loop_top() {  // gameplay.hpp:1444-1449
    if (g.frame % g.pc <= 1) g.spawn_chest();
    if (g.frame % g.pz <= 1) g.spawn_zombie_npc();
    if (g.frame % g.ph <= 1) g.spawn_human_npc();
}

int ended = 0;
void end_check() { ended = g.check_end() ? 1 : 0; }  // gameplay.hpp:1450 (Timer: the reference's clock is time(0), :1145)

long long phase_draws[6];

void half_tick(int k) {  // gameplay.hpp:1457-1463 == 1465-1471 (view / find_recom / render_it draw; `start` is the frame clock)
    long long &j = Environment::Random::jomle, j0 = j;
    g.update_tmp();
    g.hit_human(), g.hit_zombie();
    phase_draws[5] += j - j0, j0 = j;
    ++g.frame;
    g.updmap();
    g.update_bull();
    phase_draws[k] = j - j0;
}

void one_step(const std::string &cmds) {
    // chars[k] is the command of human slot k.  Agents are numbered in the order prepare() made them: the player's first
    // when the player has one; with a keyboard player (agents 0) and USE_AGENT_IN_SQUAD_NPCS, human k's agent is k - 1
    const size_t off = using_an_agent ? 0 : 1;
    for (size_t k = off; k < cmds.size(); ++k) script[(int)(k - off)] = cmds[k];
    // get_my_action, gameplay.hpp:955-958 (my_command = the keyboard: the scripted char is what was "typed")
    command[ind] = cmds.empty() ? '+' : cmds[0];
    if (using_an_agent) {
        char c = g.bot(hum[ind]);
        if (!g.manual && command[ind] != '3') command[ind] = c;
    }
    if (g.online && !g.replay_mode) client.send_it();  // gameplay.hpp:959-960
    // draws per phase (the generator's own counter, random.hpp:29): zombie_action, update_bull (1st), human_action,
    // update_bull (2nd), the next loop top, everything else
    long long &j = Environment::Random::jomle, j0 = j;
    for (auto &x : phase_draws) x = 0;
    g.zombie_action();   // gameplay.hpp:1455
    phase_draws[0] = j - j0, j0 = j;
    g.portal_damage();   // :1456
    half_tick(1);        // :1457-1463
    j0 = j;
    g.human_action();    // :1464
    phase_draws[2] = j - j0, j0 = j;
    half_tick(3);        // :1465-1471
    j0 = j;
    ++steps;
    loop_top();          // the next iteration's :1444-1449
    phase_draws[4] = j - j0;
    end_check();         // :1450
}

// The state in the word order of include/strikeforce.h's dump records (sf_human_rec 28 words with `profile` = -1: the
// reference has no such field; sf_zombie_rec 7, sf_bullet_rec 11, sf_portal_rec 4; dead zombie / bullet / exit slots as
// zeros), one line per table, then the cells: flag byte per cell as hex (SF_CELL_* bits), damage and exit number sparse.
void dump(int H_, int Z_, int B_, int P_) {
    printf("hdr %lld %lld %lld %lld %lld %lld %lld %d", g.frame, g.kills, g.teams_kills, g.loot, g.chest,
           Environment::Random::jomle, steps, ind);
    for (int i = 0; i < 18; ++i) printf(" %lld", Environment::Random::random[i]);
    printf("\nH");
    for (int i = 0; i < H_; ++i) {
        CH::Human &h = hum[i];
        auto c = h.get_cor();
        if (c.size() < 3) c = {0, 0, 0};
        printf(" %d %d %d -1 %d %d %d %d %d %d %d %d %d %d %d %d %d", (int)mh[i], (int)remote[i], (int)h.is_rnpc(), c[0],
               c[1], c[2], h.get_way(), h.get_team(), h.get_Hp(), h.get_stamina(), h.get_mindamage(), h.get_kills(),
               h.get_damage(), h.get_effect(), h.backpack.vec, h.backpack.ind);
        for (int k = 0; k < 4; ++k) printf(" %d", h.backpack.list_cons[k].second);
        for (int k = 0; k < 4; ++k) printf(" %d", h.backpack.list_throw[k].second.second);
        printf(" %d %d %d", h.backpack.get_blocks(), h.backpack.get_portals(), h.backpack.get_portal_ind());
    }
    printf("\nA");  // Human::active_agent per slot (Character.hpp:291)
    for (int i = 0; i < H_; ++i) printf(" %d", (int)hum[i].get_active_agent());
    printf("\nZ");
    for (int i = 0; i < Z_; ++i) {
        if (!mz[i]) { printf(" 0 0 0 0 0 0 0"); continue; }
        auto c = zomb[i].get_cor();
        printf(" 1 %d %d %d %d %d %d", c[0], c[1], c[2], zomb[i].get_Hp(), zomb[i].get_mindamage(), (int)zomb[i].is_super());
    }
    printf("\nB");
    for (int i = 0; i < B_; ++i) {
        if (!mb[i]) { printf(" 0 0 0 0 0 0 0 0 0 0 0"); continue; }
        auto c = bull[i].get_cor();
        auto d = bull[i].get_dcor();
        int trav = abs(c[0] - d[0]) + abs(c[1] - d[1]) + abs(c[2] - d[2]);
        uintptr_t o = bull[i].get_owner();
        int owner = o ? (int)(reinterpret_cast<CH::Human *>(o) - hum) + 1 : 0;
        const node &n = g.themap[c[0]][c[1]][c[2]];
        printf(" 1 %d %d %d %d %d %d %d %d %d %d", c[0], c[1], c[2], bull[i].get_way(), trav, bull[i].get_damage(),
               bull[i].get_effect(), bull[i].get_range(), owner, (int)(n.s[2] && n.bullet == &bull[i]));
    }
    printf("\nP");
    for (int i = 0; i < P_; ++i) {
        if (!active[i]) { printf(" 0 0 0 0"); continue; }
        printf(" 1 %d %d %d", portal[i][0], portal[i][1], portal[i][2]);
    }
    long long over = 0;  // live entities beyond the dumped slots: the caller's caps are too small for this run
    for (int i = H_; i < H; ++i) over += mh[i];
    for (int i = Z_; i < Z; ++i) over += mz[i];
    for (int i = B_; i < B; ++i) over += mb[i];
    for (int i = P_; i < B; ++i) over += active[i];
    printf("\nover %lld\nF ", over);
    std::string sparse;
    char buf[64];
    for (int f = 0; f < F; ++f)
        for (int r = 0; r < N; ++r)
            for (int c = 0; c < M; ++c) {
                const node &n = g.themap[f][r][c];
                int fl = 0;
                if (n.s[3]) fl |= 1;    // SF_CELL_WALL
                if (n.s[10]) fl |= 2;   // SF_CELL_TEMP
                if (n.s[5]) fl |= 4;    // SF_CELL_PIN_UP
                if (n.s[6]) fl |= 8;    // SF_CELL_PIN_DN
                if (n.s[7]) fl |= 16;   // SF_CELL_POUT
                if (n.s[4]) fl |= 32 | ((int)(n.cons - Environment::Item::cons) << 6);  // SF_CELL_CHEST + type
                printf("%02x", fl);
                int ci = (f * N + r) * M + c;
                if (n.s[10] && n.dmg) snprintf(buf, sizeof buf, " d%d:%d", ci, n.dmg), sparse += buf;
                if (n.s[5] || n.s[6]) snprintf(buf, sizeof buf, " x%d:%d", ci, n.portal_ind), sparse += buf;
            }
    printf("\nS%s\n", sparse.c_str());
}

void print_obs(const std::vector<float> &o) {
    printf("obs %zu\n", o.size());
    for (size_t i = 0; i < o.size(); ++i) {
        uint32_t u;
        memcpy(&u, &o[i], 4);
        if (u) printf("%zx:%x\n", i, u);
    }
}

}  // namespace

int main() {
    Environment::Item::download_items();  // main.cpp:29
    Environment::Random::make_p();        // main.cpp:30
    char line[4096];
    while (fgets(line, sizeof line, stdin)) {
        std::string s(line);
        while (!s.empty() && (s.back() == '\n' || s.back() == '\r')) s.pop_back();
        if (s.rfind("init ", 0) == 0) {
            char path[2048], mode[64];
            int level, agents;
            if (sscanf(s.c_str() + 5, "%2047s %63s %d %d", path, mode, &level, &agents) != 4) { printf("error init\nend\n"); fflush(stdout); continue; }
            user = "ref_tick";
            CH::me.build(false, "", path);  // Character.hpp:650 with an explicit file (enter.hpp:43 reads the account's)
            for (char *q = mode; *q; ++q)
                if (*q == '_') *q = ' ';
            g.mode = mode, g.level = level, g.manual = !agents;
            printf("ok dims %d %d %d %d %d %d %d\n", F, N, M, H, Z, B, C);
        } else if (s.rfind("reset ", 0) == 0) {
            long long tb, serial;
            sscanf(s.c_str() + 6, "%lld %lld", &tb, &serial);
            bool agents = !g.manual || using_an_agent;
            g.manual = !agents;
            g.chest = 0;  // gameplay.hpp:1234 does not reset it: every reset here is a first game
            g.setup();    // gameplay.hpp:1231-1277 -> load_data() :1741-1925
            g.manual = !agents;  // load_data leaves `manual = true` for the keyboard toggle ('3'); agents mode = automate
            bench_tb = tb, bench_serial = serial;
            Environment::Random::_srand(tb, serial);  // the explicit seed replaces time()/libc rand, gameplay.hpp:1233,1745-1747
            steps = 0;
            ++g.frame;   // gameplay.hpp:1441
            loop_top();
            end_check();
            printf("ok\n");
        } else if (s.rfind("logging ", 0) == 0) {
            g.enable_logging = atoi(s.c_str() + 8) != 0;
            g.log_filename.clear();
            printf("ok\n");
        } else if (s.rfind("replay ", 0) == 0) {
            g.replay_mode = atoi(s.c_str() + 7) != 0;
            printf("ok\n");
        } else if (s == "reset_native") {
            g.chest = 0;
            g.setup();  // load_data(): srand(tb) ... _srand(tb, serial_number), or the sample's header in replay mode,
                        // or (online) the match server's seed, indices, teams and the other players' records
            if (g.online && !g.replay_mode && !disconnect) client.prepare();  // gameplay.hpp:1432-1433
            steps = 0;
            ++g.frame;
            loop_top();
            end_check();
            if (g.online)
                printf("ok %lld %lld %d %d %d %d\n", client.tb, g.serial_number, ind, client.n, client.team, (int)disconnect);
            else
                printf("ok %lld %lld\n", (long long)g.tb, g.serial_number);
        } else if (s == "rivals") {
            printf("ok %d\n", (int)g.rivals_are_dead());
        } else if (s == "logclose") {
            g.log_file.close();
            printf("ok %s\n", g.log_filename.c_str());
        } else if (s.rfind("step", 0) == 0) {
            one_step(s.size() > 5 ? s.substr(5) : std::string());
            printf("ok %lld %lld %lld %lld %lld %lld %d\n", phase_draws[0], phase_draws[1], phase_draws[2], phase_draws[3],
                   phase_draws[4], phase_draws[5], ended);
        } else if (s.rfind("dump ", 0) == 0) {
            int a, b, c, d;
            sscanf(s.c_str() + 5, "%d %d %d %d", &a, &b, &c, &d);
            dump(a, b, c, d);
        } else if (s == "calls") {
            for (auto &c : calls) printf("call %d %c %d %d %lld\n", c.id, c.kind, c.a, c.b, c.frame);
            calls.clear();
        } else if (s.rfind("obs ", 0) == 0) {
            print_obs(last_obs[atoi(s.c_str() + 4)]);
        } else if (s.rfind("observe ", 0) == 0) {
            // gameplay::bot (Custom.hpp:137-159) on hum[slot], whoever it is: a probe agent receives the vector
            int slot = atoi(s.c_str() + 8);
            CH::Human &h = hum[slot];
            bool had = h.get_active_agent();
            Agent *old = h.agent;
            probing = true;
            Agent probe;
            h.agent = &probe;
            h.set_agent_active();
            g.bot(h);
            h.agent = old;
            if (!had) h.reset_agent_active();
            probing = false;
            print_obs(probe_obs);
        } else if (s.rfind("bench ", 0) == 0) {
            long long n = 0, done_steps = 0, resets = 0;
            unsigned x = 0;
            sscanf(s.c_str() + 6, "%lld %u", &n, &x);
            static const char table[] = "+qeuzxawsdfghjkl;'cvbnm,./[]";
            long long tb = bench_tb;
            auto t0 = std::chrono::steady_clock::now();
            for (; done_steps < n; ++done_steps) {
                if (hum[ind].get_Hp() <= 0 || !mh[ind]) {  // "You Died": play() returns, the next game starts
                    g.chest = 0;
                    g.setup();
                    Environment::Random::_srand(++tb, bench_serial);
                    ++g.frame;
                    loop_top();
                    ++resets;
                }
                x = x * 1664525u + 1013904223u;
                one_step(std::string(1, table[(x >> 16) % 28]));
            }
            double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            bench_tb = tb;
            printf("ok %lld %.6f %lld\n", done_steps, sec, resets);
        } else if (s == "quit") {
            break;
        } else {
            printf("error unknown command\n");
        }
        printf("end\n");
        fflush(stdout);
    }
    return 0;
}
