// examples/replay_sample.cpp — a game the reference client LOGGED (`.sf_sample`, gameplay.hpp:1784-1794,966-967) replayed
// through the C-ABI on the reference's own world (map/floor1-3.txt, 3 x 30 x 100), with include/sf_sample.hpp.
//
//   g++ -std=c++17 -O2 -I include examples/replay_sample.cpp -L strikeforce_amd -lstrikeforce_amd
//       -Wl,-rpath,$PWD/strikeforce_amd -o /tmp/replay_sample
//   /tmp/replay_sample tests/golden/maps game.sf_sample <mode 0 Solo | 1 Timer | 2 Squad> <level> [copy.sf_sample]
//
// Prints the sample's header, the number of iterations played and the final state digest; with a fifth argument the
// sample is written out again (the reference's byte layout: the copy replays in the reference itself).
#include <cstdio>
#include <cstdlib>

#include "sf_sample.hpp"

int main(int argc, char **argv) {
  if (argc < 5) return fprintf(stderr, "usage: %s <maps dir> <sample> <mode> <level> [copy]\n", argv[0]), 2;
  sf::Sample s;
  std::string why;
  if (!sf::read_sample(argv[2], s, &why)) return fprintf(stderr, "%s: %s\n", argv[2], why.c_str()), 1;
  sf_config cfg;
  sf_config_defaults(&cfg);
  cfg.arenas = 1, cfg.floors = 3, cfg.rows = 30, cfg.cols = 100;  // gameplay.hpp:37
  // pools for a whole game of the reference (its own hold 9000): 1024 zombies and 512 exits live in the arena's LDS
  cfg.cap_humans = 64, cfg.cap_zombies = 1024, cfg.cap_bullets = 256, cfg.cap_portals = 512, cfg.cap_chests = 9000;
  cfg.mode = atoi(argv[3]), cfg.level = atoi(argv[4]), cfg.n_agents = 1, cfg.auto_reset = 0;
  cfg.timer_frames_per_level = 1 << 20;
  cfg.player = sf::profile_of(s.record);  // the record comes from the file (Human::scan_file, Character.hpp:570-617)
  std::string chars;
  std::vector<int16_t> portal;
  if (!sf::load_reference_maps(argv[1], 3, 30, 100, chars, portal)) return fprintf(stderr, "cannot read the maps in %s\n", argv[1]), 1;
  cfg.map = chars.data(), cfg.map_portal = portal.data();
  sf_env *env = nullptr;
  if (sf_create(&cfg, &env) != SF_OK) return fprintf(stderr, "sf_create: %s\n", sf_last_error()), 1;
  const long n = sf::replay(env, s);
  if (n < 0) return fprintf(stderr, "replay: %s\n", sf_last_error()), 1;
  uint64_t digest = 0;
  sf_state_digest(env, &digest);
  printf("sample tb %llu serial %llu ind %d team %d name %s commands %zu\n", (unsigned long long)s.tb, (unsigned long long)s.serial, s.ind,
         s.team, s.name.c_str(), s.commands.size());
  printf("iterations %ld\ndigest %016llx\n", n, (unsigned long long)digest);
  if (argc > 5 && !sf::write_sample(argv[5], s)) return fprintf(stderr, "cannot write %s\n", argv[5]), 1;
  sf_destroy(env);
  return 0;
}
