// examples/match_client.cpp — one player of a lock-step Battle match (the reference's online mode) with an sf_env as
// its world, through the reference's own match server (StrikeForce-server/server.cpp), with include/sf_lockstep.hpp.
//
//   g++ -std=c++17 -O2 -I include examples/match_client.cpp -L strikeforce_amd -lstrikeforce_amd
//       -Wl,-rpath,$PWD/strikeforce_amd -o /tmp/match_client
//   /tmp/match_client <host> <port> <password> <name> <world file> <iterations> <policy seed> <record: 32 integers>
//
// world file: "rows cols H Z B P", then rows*cols map characters (# . O ^ v, whitespace ignored), then rows*cols exit
// numbers (the entry of a '^' / 'v' cell; -1 elsewhere).  The player acts at random over the 28 commands of the bench's
// random-action agent.  Prints the match as this client saw it: "match tb serial n ind team", then per iteration
// "it <k> <the n command chars as stepped> <state digest>", then "end <iterations> <won|died|quit>".
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "sf_lockstep.hpp"

int main(int argc, char **argv) {
  if (argc < 8 + 32) return fprintf(stderr, "usage: %s host port password name world iterations seed record[32]\n", argv[0]), 2;
  int32_t record[32];
  for (int i = 0; i < 32; ++i) record[i] = atoi(argv[8 + i]);
  std::ifstream wf(argv[5]);
  int rows, cols, H, Z, B, P;
  if (!(wf >> rows >> cols >> H >> Z >> B >> P)) return fprintf(stderr, "bad world file\n"), 1;
  std::string chars;
  std::vector<int16_t> portal;
  for (int i = 0; i < rows * cols; ++i) {
    char c;
    wf >> c;
    chars.push_back(c);
  }
  for (int i = 0; i < rows * cols; ++i) {
    int v;
    wf >> v;
    portal.push_back((int16_t)v);
  }
  if (!wf) return fprintf(stderr, "world file too short\n"), 1;
  sf::MatchClient client;
  if (!client.connect_to(argv[1], atoi(argv[2]), argv[3], argv[4], record)) return fprintf(stderr, "connect: %s\n", client.error.c_str()), 1;
  printf("match %llu %llu %d %d %d\n", (unsigned long long)client.tb, (unsigned long long)client.serial, client.n, client.ind, client.team);
  sf_config cfg;
  sf_config_defaults(&cfg);
  cfg.floors = 1, cfg.rows = rows, cfg.cols = cols;
  cfg.cap_humans = H, cfg.cap_zombies = Z, cfg.cap_bullets = B, cfg.cap_portals = P, cfg.cap_chests = 9000;
  cfg.map = chars.data(), cfg.map_portal = portal.data();
  client.configure(cfg);
  sf_env *env = nullptr;
  if (sf_create(&cfg, &env) != SF_OK) return fprintf(stderr, "sf_create: %s\n", sf_last_error()), 1;
  static const char COMMANDS[] = "+qeuzxawsdfghjkl;'cvbnm,./[]";  // the 30 codes minus '3' and '_' 
  uint32_t x = (uint32_t)atoi(argv[7]);
  auto policy = [&](long) {
    x = x * 1664525u + 1013904223u;
    return COMMANDS[(x >> 16) % 28u];
  };
  std::string how;
  const long n = sf::play_match(client, env, policy, atol(argv[6]), &how, [&](long it, const std::vector<uint8_t> &cmd) {
    uint64_t d = 0;
    sf_state_digest(env, &d);
    printf("it %ld %s %016llx\n", it, std::string(cmd.begin(), cmd.end()).c_str(), (unsigned long long)d);
  });
  printf("end %ld %s\n", n, how.c_str());
  sf_destroy(env);
  return n < 0;
}
