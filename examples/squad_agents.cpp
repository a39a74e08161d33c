// examples/squad_agents.cpp — ten reference-style Agents in one Squad game (the reference's USE_AGENT_IN_SQUAD_NPCS
// build, gameplay.hpp:1883-1885,1896-1898), driven through the C-ABI by include/sf_agent_adapter.hpp on the reference's
// own world (map/floor1-3.txt, 3 x 30 x 100).
//
//   g++ -std=c++17 -O2 -I include examples/squad_agents.cpp -L strikeforce_amd -lstrikeforce_amd
//       -Wl,-rpath,$PWD/strikeforce_amd -o /tmp/squad_agents ; /tmp/squad_agents tests/golden/maps 300
//
// Every Agent logs what the simulator asks of it, in order: "P id action" (predict), "U id action imitate" (update),
// "D id" (destroyed: its human died), "S step" between iterations.  The policy is a fixed function of (agent, step) so
// that the same game can be scripted into the reference itself: tests/test_squad_agents_example.py holds this log
// against the reference's own call sequence, and the final digest against the oracle's.
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "sf_agent_adapter.hpp"

static int g_next_id = 0, g_step = 0;

class Agent {
  int id;

 public:
  Agent() : id(g_next_id++) { printf("N %d\n", id); }
  ~Agent() { printf("D %d\n", id); }
  int predict(const std::vector<float> &obs) {
    const int act = (id * 7 + g_step * 3 + g_step / 5) % 9;
    printf("P %d %d %d\n", id, act, (int)obs.size());
    return act;
  }
  void update(int action, bool imitate) { printf("U %d %d %d\n", id, action, (int)imitate); }
  bool in_training() { return false; }
  bool is_manual() { return false; }
};

// map/floorK.txt as gameplay::setup() reads it (gameplay.hpp:1249-1274): `f >> c`, '^' / 'v' followed by an exit number
static void read_floor(const std::string &path, int rows, int cols, std::string &chars, std::vector<int16_t> &portal) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open " + path);
  for (int i = 0; i < rows * cols; ++i) {
    char c;
    int idx = -1;
    f >> c;
    if (c == '^' || c == 'v')
      f >> idx;
    else if (c != '#' && c != 'O')
      c = '.';
    chars.push_back(c);
    portal.push_back((int16_t)idx);
  }
}

int main(int argc, char **argv) {
  const std::string dir = argc > 1 ? argv[1] : "tests/golden/maps";
  const int steps = argc > 2 ? atoi(argv[2]) : 300;
  sf_config cfg;
  sf_config_defaults(&cfg);
  cfg.arenas = 1, cfg.floors = 3, cfg.rows = 30, cfg.cols = 100;  // gameplay.hpp:37
  // pools for a whole game of the reference (its own hold 9000): 1024 zombies and 512 exits live in the arena's LDS
  cfg.cap_humans = 64, cfg.cap_zombies = 1024, cfg.cap_bullets = 256, cfg.cap_portals = 512, cfg.cap_chests = 9000;
  cfg.mode = SF_MODE_SQUAD, cfg.level = 2, cfg.n_agents = 10, cfg.auto_reset = 0;
  cfg.timer_frames_per_level = 1 << 20;
  // the reference's level-10 account (accounts/game/1): 15000 Hp, 1000 mindamage, one of everything
  const int32_t rich[32] = {15000, 1000, 15000, 10, 10, 10, 300000, 60, 0, 0, 0, 1, 1, 1, 34, 1, 1, 1, 1, 1, 1, 1, 1,
                            1, 1, 1, 1, 1, 1, 1, 1, 56};
  sf_profile &p = cfg.player;
  p.def_hp = rich[0], p.mindamage_def = rich[1], p.def_stamina = rich[2];
  p.level_solo = rich[3], p.level_timer = rich[4], p.level_squad = rich[5], p.money = rich[6];
  p.rate_solo = rich[7], p.rate_timer = rich[8], p.rate_squad = rich[9], p.rate = rich[10];
  for (int i = 0; i < 4; ++i) p.cons[i] = rich[11 + i];
  for (int i = 0; i < 4; ++i) p.throw_lvl_cnt[i][0] = rich[15 + 2 * i], p.throw_lvl_cnt[i][1] = rich[16 + 2 * i];
  for (int i = 0; i < 8; ++i) p.weapon_lvl[i] = rich[23 + i];
  p.backpack_lvl = rich[31];
  try {
    std::string chars;
    std::vector<int16_t> portal;
    for (int k = 1; k <= 3; ++k) read_floor(dir + "/floor" + std::to_string(k) + ".txt", 30, 100, chars, portal);
    cfg.map = chars.data(), cfg.map_portal = portal.data();
    sf::AgentRunner<Agent> run(cfg, "+xzqeawsd");  // the action string of bots/bot-0.5/Custom.hpp:162
    const uint64_t tb = 1700000000ull, serial = 123456789ull;
    run.reset(&tb, &serial);
    for (g_step = 0; g_step < steps; ++g_step) {
      printf("S %d\n", g_step);
      if (run.step()) break;
    }
    uint64_t dig = 0;
    if (sf_state_digest(run.env(), &dig) != SF_OK) throw std::runtime_error(sf_last_error());
    printf("E %d\n", g_step);
    printf("digest %016llx\n", (unsigned long long)dig);
  } catch (const std::exception &e) {
    fprintf(stderr, "squad_agents: %s\n", e.what());
    return 1;
  }
  return 0;
}
