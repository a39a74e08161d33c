// sf_sample.hpp — the reference's `.sf_sample` trajectory format for C++ hosts of the C-ABI (header-only).
//
// A `.sf_sample` file is what the reference client writes while a game is played with logging on
// (StrikeForce-client/gameplay.hpp:1784-1794 opens it and writes the header, :1910-1914 the player's blob,
// :966-967 one command per loop iteration) and reads back in replay mode (:1749-1782, :968-969).  The player's blob is
// Human::log_file (Character.hpp:619-648), read by Human::scan_file (:570-617): the name and the 32 integers of the
// character record, one per line.  Offline (Solo / Timer / Squad) files look like
//
//     <tb> <serial>
//     1 <ind> <team>
//     <name>
//     <def_Hp> ... 31 more integers, one per line
//     <one command char per loop iteration, one per line>
//
// sf::read_sample / sf::write_sample read and write that layout (a file written here replays in the reference, one
// the reference logged replays here: tests/test_cpp_sample.py does both against the reference's own build);
// sf::replay feeds a sample to an sf_env through sf_reset / sf_step / sf_done.  Python twin: strikeforce_amd/replay.py.
#ifndef SF_SAMPLE_HPP
#define SF_SAMPLE_HPP

#include <stdint.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "strikeforce.h"

namespace sf {

struct Sample {
  uint64_t tb = 0, serial = 0;  // the seed pair of the game: gameplay.hpp:1745-1746 (`_srand(tb, serial_number)`)
  int players = 1, ind = 0, team = 1;
  std::string name = "player";
  int32_t record[32] = {0};     // the character record, Character.hpp:669-689 (33 tokens minus the name)
  std::string commands;         // one reference command char per loop iteration (gameplay.hpp:45 + '_')
};

// sf_profile is those 32 integers in file order (include/strikeforce.h)
static_assert(sizeof(sf_profile) == 32 * sizeof(int32_t), "sf_profile = the 32 integers of a character record");
inline sf_profile profile_of(const int32_t record[32]) {
  sf_profile p;
  memcpy(&p, record, sizeof p);
  return p;
}

// Parses the way the reference does: whitespace-separated tokens (`operator>>`), commands one char at a time
// (`replay_file >> command[ind]` reads a char: a token "ab" is two commands).  Returns false (and says why) on a file
// that is not an offline sample.
inline bool read_sample(const std::string &path, Sample &s, std::string *why = nullptr) {
  auto fail = [&](const char *m) {
    if (why) *why = m;
    return false;
  };
  std::ifstream f(path.c_str());
  if (!f) return fail("cannot open the file");
  long long tb, serial;
  if (!(f >> tb >> serial >> s.players >> s.ind >> s.team)) return fail("header: expected `tb serial` and `players ind team`");
  if (s.players != 1) return fail("only offline samples (one player) are supported");
  s.tb = (uint64_t)tb, s.serial = (uint64_t)serial;
  if (!(f >> s.name)) return fail("player name missing");
  for (int i = 0; i < 32; ++i)
    if (!(f >> s.record[i])) return fail("character record: fewer than 32 integers");
  s.commands.clear();
  char c;
  while (f >> c) s.commands.push_back(c);
  return true;
}

// The reference logger's byte layout: header, Human::log_file blob, then `command << '\n'` per iteration.
inline bool write_sample(const std::string &path, const Sample &s) {
  std::ofstream f(path.c_str());
  if (!f) return false;
  f << s.tb << ' ' << s.serial << '\n' << 1 << ' ' << s.ind << ' ' << s.team << '\n' << s.name << '\n';
  for (int i = 0; i < 32; ++i) f << s.record[i] << '\n';
  for (char c : s.commands) f << c << '\n';
  return (bool)f;
}

// map/floor1.txt .. floor<floors>.txt of a reference checkout, read the way gameplay::setup() does
// (gameplay.hpp:1249-1274: `f >> c`, a '^' or 'v' is followed by the number of the exit it leads to; any character other
// than # . O ^ v is an empty cell): fills what sf_config::map / map_portal point at.
inline bool load_reference_maps(const std::string &dir, int floors, int rows, int cols, std::string &chars,
                                std::vector<int16_t> &portal) {
  chars.clear(), portal.clear();
  for (int k = 1; k <= floors; ++k) {
    std::ifstream f((dir + "/floor" + std::to_string(k) + ".txt").c_str());
    if (!f) return false;
    for (int i = 0; i < rows * cols; ++i) {
      char c;
      int idx = -1;
      if (!(f >> c)) return false;
      if (c == '^' || c == 'v') {
        if (!(f >> idx)) return false;
      } else if (c != '#' && c != 'O') {
        c = '.';
      }
      chars.push_back(c);
      portal.push_back((int16_t)idx);
    }
  }
  return true;
}

// Replays the sample on `env` — created for ONE arena with cfg.player = profile_of(sample.record), auto_reset off and
// the mode / level / map of the logged game — the way the reference's replay loop does: one command per iteration
// until the game ends (gameplay.hpp:1450 `if(check_end()) break;`) or the commands run out.  Returns the number of
// iterations played, or a negative SF_ERR_* code.
inline long replay(sf_env *env, const Sample &s) {
  const uint64_t tb = s.tb, serial = s.serial;
  int rc = sf_reset(env, &tb, &serial);
  if (rc != SF_OK) return rc;
  long n = 0;
  for (char c : s.commands) {
    uint8_t done = 0;
    if ((rc = sf_done(env, &done)) != SF_OK) return rc;
    if (done) break;
    const uint8_t cmd = (uint8_t)c;
    if ((rc = sf_step(env, &cmd)) != SF_OK) return rc;
    ++n;
  }
  return n;
}

}  // namespace sf
#endif  // SF_SAMPLE_HPP
