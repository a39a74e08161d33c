// examples/drive_agent.cpp — a reference-style Agent driven through the C-ABI with include/sf_agent_adapter.hpp.
//
//   g++ -std=c++17 -O2 -I include examples/drive_agent.cpp -L strikeforce_amd -lstrikeforce_amd
//       -Wl,-rpath,$PWD/strikeforce_amd -o /tmp/drive_agent ; /tmp/drive_agent
//
// The Agent below has exactly the member surface of the reference's bots (bots/bot-0.5/Agent.hpp:178,217,239,266);
// its "policy" is a fixed cycle over the 9-action set "+xzqeawsd" of bots/bot-0.5/Custom.hpp:162 so that the program
// is self-contained (the shipped bots need libtorch).  Prints the per-arena state digests after the run; the same
// digests come out of the Python path (tests/test_cpp_example.py checks that).
#include <cstdio>
#include <string>
#include <vector>

#include "sf_agent_adapter.hpp"

class Agent {
  int t = 0;

 public:
  int predict(const std::vector<float> &obs) {
    // look at the centre cell's "wall ahead" style features just to touch the observation
    const float me = obs[0 * 961 + 15 * 31 + 15];
    return (t++ + (me > 0.f ? 1 : 0)) % 9;
  }
  void update(int, bool) {}
  bool in_training() { return false; }
  bool is_manual() { return false; }
};

int main(int argc, char **argv) {
  const int arenas = argc > 1 ? atoi(argv[1]) : 8, steps = argc > 2 ? atoi(argv[2]) : 200;
  sf_config cfg;
  sf_config_defaults(&cfg);
  cfg.arenas = arenas, cfg.floors = 1, cfg.rows = 32, cfg.cols = 32;
  cfg.cap_humans = 4, cfg.cap_zombies = 8, cfg.cap_bullets = 16, cfg.cap_portals = 4;
  cfg.mode = SF_MODE_SOLO, cfg.level = 1, cfg.n_agents = 1, cfg.auto_reset = 1;
  cfg.player = cfg.npc;  // the armed profile (character/human_enemy.txt)
  std::string map((size_t)32 * 32, '.');
  for (int i = 0; i < 32; ++i) map[i] = map[31 * 32 + i] = map[i * 32] = map[i * 32 + 31] = '#';
  for (int r = 4; r < 28; r += 5)
    for (int c = 3; c < 29; c += 7) map[r * 32 + c] = '#';
  cfg.map = map.data();
  try {
    sf::AgentRunner<Agent> run(cfg, "+xzqeawsd");
    std::vector<uint64_t> tb(arenas), serial(arenas, 123456789ull);
    for (int a = 0; a < arenas; ++a) tb[a] = 1700000000ull + (uint64_t)a;
    run.reset(tb.data(), serial.data());
    int ended = 0;
    for (int s = 0; s < steps; ++s) ended += run.step();
    std::vector<uint64_t> dig(arenas);
    if (sf_state_digest(run.env(), dig.data()) != SF_OK) throw std::runtime_error(sf_last_error());
    printf("episodes_ended %d\n", ended);
    for (int a = 0; a < arenas; ++a) printf("digest %d %llu\n", a, (unsigned long long)dig[a]);
  } catch (const std::exception &e) {
    fprintf(stderr, "drive_agent: %s\n", e.what());
    return 1;
  }
  return 0;
}
