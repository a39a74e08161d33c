// sf_lockstep.hpp — the client side of the reference's lock-step match protocol for C++ hosts of the C-ABI
// (header-only, POSIX sockets).  Python twin: strikeforce_amd/lockstep.py.
//
// The reference's online mode has no authoritative simulation: the match server (StrikeForce-server/server.cpp) hands
// every client the same seed and its player index, forwards the account blobs, and then only relays one command char
// per player per loop iteration; every client simulates the whole world (StrikeForce-client/gameplay.hpp:66-193,
// 1795-1859).  sf::MatchClient speaks that protocol; sf::play_match runs gameplay::play()'s loop with an sf_env as the
// world, so that a batch simulator on an MI355X can sit in a match beside reference clients.
//
// Wire format (all strings NUL-terminated, read byte-wise like basic.hpp:261-271 my_recv):
//   -> password                           <- "A" | "R"                    Client::start   gameplay.hpp:66-111
//   <- "<tb> <serial>"    <- "<n> <ind> <team>"                           server.cpp:239-246
//   -> own account blob ("name\nHp\n..."), <- for every other player in index order: blob, team   gameplay.hpp:120-151
//   per iteration: -> own command char;  <- the command of every other player whose client was alive when the
//   iteration began, in index order       send_it / recieve gameplay.hpp:113-118,170-193; server.cpp:76-117
//   leaving: '~' when the own player is dead, '+' when all rivals are dead, '_' to quit   gameplay.hpp:1102-1143,939-953
#ifndef SF_LOCKSTEP_HPP
#define SF_LOCKSTEP_HPP

#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <stdint.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include <functional>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "strikeforce.h"

namespace sf {

class MatchClient {
 public:
  uint64_t tb = 0, serial = 0;
  int n = 0, ind = 0, team = 0;
  std::vector<int> teams;                      // every player's team (server.cpp:239-246)
  std::vector<std::vector<int32_t>> records;   // every player's character record, 32 integers (gameplay.hpp:120-151)
  std::string error;

  ~MatchClient() { close_socket(); }

  // Client::start + give_info + get_info (gameplay.hpp:66-151): joins the match and blocks until the server has all
  // its players.  `record`: the 32 integers of this player's character record.
  bool connect_to(const std::string &host, int port, const std::string &password, const std::string &name,
                  const int32_t record[32]) {
    fd_ = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd_ < 0) return fail("socket()");
    sockaddr_in a;
    memset(&a, 0, sizeof a);
    a.sin_family = AF_INET, a.sin_port = htons((uint16_t)port);
    if (inet_pton(AF_INET, host.c_str(), &a.sin_addr) != 1) return fail("bad IPv4 address");
    if (::connect(fd_, (sockaddr *)&a, sizeof a) != 0) return fail("connect()");
    int one = 1;
    setsockopt(fd_, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    std::string s;
    if (!send_cstr(password) || !recv_cstr(s)) return false;
    if (s.empty() || s[0] != 'A') return fail("the server refused the password");
    long long t0, t1;
    if (!recv_cstr(s) || !(std::istringstream(s) >> t0 >> t1)) return fail("seed line");
    tb = (uint64_t)t0, serial = (uint64_t)t1;
    if (!recv_cstr(s) || !(std::istringstream(s) >> n >> ind >> team) || n < 1 || ind < 0 || ind >= n || n > SF_MAX_AGENTS)
      return fail("player line");
    std::ostringstream blob;  // the account file of give_info(): 33 tokens, one per line
    blob << name;
    for (int i = 0; i < 32; ++i) blob << '\n' << record[i];
    if (!send_cstr(blob.str())) return false;
    teams.assign((size_t)n, 0), records.assign((size_t)n, std::vector<int32_t>());
    teams[(size_t)ind] = team, records[(size_t)ind].assign(record, record + 32);
    for (int i = 0; i < n; ++i) {
      if (i == ind) continue;
      if (!recv_cstr(s)) return false;
      std::istringstream is(s);
      std::string nm;
      is >> nm;
      std::vector<int32_t> r(32);
      for (int k = 0; k < 32; ++k)
        if (!(is >> r[(size_t)k])) return fail("an account blob has a name and 32 integers");
      records[(size_t)i] = r;
      if (!recv_cstr(s)) return false;
      teams[(size_t)i] = atoi(s.c_str());
    }
    return true;
  }

  // The one-arena Battle configuration of this match as this client sees it (`ind` = the server-assigned index): mode,
  // players, teams and every player's record.  The caller has filled the world (dims, map, pools, items) before.
  void configure(sf_config &cfg) const {
    cfg.arenas = 1, cfg.mode = SF_MODE_BATTLE, cfg.level = 1, cfg.n_agents = n, cfg.ind = ind, cfg.auto_reset = 0;
    cfg.n_agent_profiles = n;
    for (int i = 0; i < n; ++i) {
      cfg.agent_team[i] = teams[(size_t)i];
      memcpy(&cfg.agent_profile[i], records[(size_t)i].data(), sizeof(sf_profile));
    }
    memcpy(&cfg.player, records[(size_t)ind].data(), sizeof(sf_profile));
  }

  bool send_command(char c) {  // Client::send_it gameplay.hpp:113-118
    const char b[2] = {c, 0};
    return send_all(b, 2);
  }
  // one command for every player index in `expected` (ascending), as the server relays them (Client::recieve :170-193)
  bool recv_commands(const std::vector<int> &expected, std::vector<uint8_t> &cmd) {
    std::string s;
    for (int i : expected) {
      if (!recv_cstr(s)) return false;
      cmd[(size_t)i] = s.empty() ? (uint8_t)'+' : (uint8_t)s[0];
    }
    return true;
  }
  void close_socket() {
    if (fd_ >= 0) ::close(fd_);
    fd_ = -1;
  }

 private:
  int fd_ = -1;
  bool fail(const char *m) {
    error = m;
    close_socket();
    return false;
  }
  bool send_all(const char *p, size_t len) {
    while (len) {
      const ssize_t k = ::send(fd_, p, len, MSG_NOSIGNAL);
      if (k <= 0) return fail("send()");
      p += k, len -= (size_t)k;
    }
    return true;
  }
  bool send_cstr(const std::string &s) { return send_all(s.c_str(), s.size() + 1); }
  bool recv_cstr(std::string &out) {
    out.clear();
    for (;;) {
      char c;
      const ssize_t k = ::recv(fd_, &c, 1, 0);
      if (k <= 0) return fail("the server closed the connection");
      if (c == 0) return true;
      out.push_back(c);
    }
  }
};

// gameplay::play()'s loop for one client of a match (gameplay.hpp:1428-1505 with `online`): `env` was created on a
// configuration that client.configure() completed; `policy(iteration)` returns this client's command char.
// Returns the number of iterations played; `how` = "won" | "died" | "quit" | an error.
inline long play_match(MatchClient &client, sf_env *env, const std::function<char(long)> &policy, long max_iterations,
                       std::string *how = nullptr, const std::function<void(long, const std::vector<uint8_t> &)> &on_step = nullptr) {
  auto say = [&](const char *s) {
    if (how) *how = s;
  };
  const uint64_t tb = client.tb, serial = client.serial;
  if (sf_reset(env, &tb, &serial) != SF_OK) return say(sf_last_error()), -1;
  const int n = client.n, ind = client.ind;
  std::set<int> quit_seen;
  std::vector<sf_human_rec> humans((size_t)SF_MAX_HUMANS);
  for (long it = 0; it < max_iterations; ++it) {
    if (sf_dump_arena(env, 0, nullptr, humans.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != SF_OK)
      return say(sf_last_error()), -1;
    uint8_t done = 0;
    if (sf_done(env, &done) != SF_OK) return say(sf_last_error()), -1;
    if (done) {  // check_end(): all rivals dead -> '+', own player dead -> '~'   gameplay.hpp:1103-1143
      const bool won = humans[(size_t)ind].alive != 0;
      client.send_command(won ? '+' : '~');
      client.close_socket();
      return say(won ? "won" : "died"), it;
    }
    const char mine = policy(it);
    if (!client.send_command(mine)) return say(client.error.c_str()), -1;
    if (mine == '_') {  // quit: the server tells the others and drops this client   gameplay.hpp:939-953
      client.close_socket();
      return say("quit"), it;
    }
    std::vector<int> expected;
    for (int i = 0; i < n; ++i)
      if (i != ind && humans[(size_t)i].alive && !quit_seen.count(i)) expected.push_back(i);
    std::vector<uint8_t> cmd((size_t)n, (uint8_t)'+');
    cmd[(size_t)ind] = (uint8_t)mine;
    if (!client.recv_commands(expected, cmd)) return say(client.error.c_str()), -1;
    for (int i : expected)
      if (cmd[(size_t)i] == '_') quit_seen.insert(i);  // its client is gone; '_' kills the player in every simulation (:696-699)
    if (sf_step(env, cmd.data()) != SF_OK) return say(sf_last_error()), -1;
    if (on_step) on_step(it, cmd);
  }
  client.send_command('_');
  client.close_socket();
  return say("quit"), max_iterations;
}

}  // namespace sf
#endif  // SF_LOCKSTEP_HPP
