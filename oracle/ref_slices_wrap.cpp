// ref_slices_wrap.cpp — TEST INFRASTRUCTURE: C entry points around the parts of the reference that compile here.
//
// oracle/ref_slices.py cuts these line ranges out of the checkout under /root/reference (by line number, with an
// anchor-text check on the first and last line of every range, into a temporary directory that is deleted after
// the build) and compiles THIS file against them with plain g++ and standard headers only:
//
//   slice_random.inc     StrikeForce-client/random.hpp:27-77      namespace Environment::Random  (make_p, binpow,
//                                                                  _rand, _srand)
//   slice_item.inc       StrikeForce-client/Item.hpp:27-194       namespace Environment::Item    (Item, Weapon, Bullet,
//                                                                  tables, download_items)
//   slice_char_cd.inc    StrikeForce-client/Character.hpp:29-47   compute_damage, wdx/wdy
//   slice_char_base.inc  StrikeForce-client/Character.hpp:225-287 class Character (hit)
//   slice_char_zomb.inc  StrikeForce-client/Character.hpp:832-871 class Zombie (punch, gen_npc), gen_zombie
//
// The sliced text is compiled as it lies: nothing is edited, no reference header is replaced by a stand-in (the
// ranges are the ones that need nothing but the standard library; Backpack/Human reach basic.hpp's terminal layer and
// class Agent, gameplay.hpp reaches SFML, so they stay out — DESIGN.md §2).  The three Character.hpp ranges sit
// inside `namespace Environment::Character{` in the reference (Character.hpp:27), which this file reopens around
// them.  Output: oracle/_ref/libsf_refslice.so (git-ignored, never committed; the slices themselves are not kept).
// Only tests/ and tests/golden/make_kat.py load it, to pin oracle/sf_oracle.c's restatement of the same functions.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <string>
#include <vector>

#include <unistd.h>

#include "slice_random.inc"
#include "slice_item.inc"
namespace Environment::Character {
#include "slice_char_cd.inc"
#include "slice_char_base.inc"
#include "slice_char_zomb.inc"
}  // namespace Environment::Character

namespace R = Environment::Random;
namespace I = Environment::Item;
namespace Ch = Environment::Character;

extern "C" {

// random.hpp:33-40 once, then _srand(tb, u_s) (random.hpp:64-76), n draws of _rand() (random.hpp:54-62) into out
// (may be null), and the generator's state random[0..17], jomle into state19 (may be null)
void ref_rand(long long tb, long long u_s, int n, int *out, long long *state19) {
  static bool made = false;
  if (!made) R::make_p(), made = true;
  R::_srand(tb, u_s);
  for (int i = 0; i < n; ++i) {
    const int v = R::_rand();
    if (out) out[i] = v;
  }
  if (state19) {
    for (int i = 0; i < 18; ++i) state19[i] = R::random[i];
    state19[18] = R::jomle;
  }
}

int ref_compute_damage(int x, int y) { return Ch::compute_damage(x, y); }

// download_items() (Item.hpp:179-188) reads ./Items/*.txt: run it with the reference's client directory as cwd.
// out: cons[4] x {price, vol, lvl, stamina, Hp, effect}, throw_[4] and w[8] x {price, vol, lvl, stamina, damage,
// effect, range} = 24 + 28 + 56 ints.  Returns 0, or -1 if the directory cannot be entered.
int ref_items(const char *client_dir, int *out) {
  char old[4096];
  if (!getcwd(old, sizeof old) || chdir(client_dir) != 0) return -1;
  I::download_items();
  if (chdir(old) != 0) return -1;
  int k = 0;
  for (int i = 0; i < 4; ++i) {
    const I::ConsumableItem &c = I::cons[i];
    out[k++] = c.get_price(), out[k++] = c.get_vol(), out[k++] = c.get_level(), out[k++] = c.get_stamina();
    out[k++] = c.get_Hp(), out[k++] = c.get_effect();
  }
  for (int i = 0; i < 4; ++i) {
    const I::Bullet &b = I::throw_[i];
    out[k++] = b.get_price(), out[k++] = b.get_vol(), out[k++] = b.get_level(), out[k++] = b.get_stamina();
    out[k++] = b.get_damage(), out[k++] = b.get_effect(), out[k++] = b.get_range();
  }
  for (int i = 0; i < 8; ++i) {
    const I::Weapon &w = I::w[i];
    out[k++] = w.get_price(), out[k++] = w.get_vol(), out[k++] = w.get_level(), out[k++] = w.get_stamina();
    out[k++] = w.get_damage(), out[k++] = w.get_effect(), out[k++] = w.get_range();
  }
  return 0;
}

// Weapon::ready + Bullet::shot at cor0 (Item.hpp:63-68,156-163), the bullet then moved to cor1 (set_cor):
// out = {damage, effect, range, way, owner, expire()} (Item.hpp:165-168)
void ref_bullet(const int *cor0, const int *cor1, int way, int damage, int effect, int range, int owner, int *out6) {
  I::Weapon w;
  w.ready(damage, effect, range);
  I::Bullet b;
  b.shot({cor0[0], cor0[1], cor0[2]}, way, w, (uintptr_t)owner);
  b.set_cor({cor1[0], cor1[1], cor1[2]});
  out6[0] = b.get_damage(), out6[1] = b.get_effect(), out6[2] = b.get_range(), out6[3] = b.get_way();
  out6[4] = (int)b.get_owner(), out6[5] = b.expire() ? 1 : 0;
}

// Character::hit (Character.hpp:242-246): out = {Hp, mindamage}
void ref_character_hit(int hp, int mindamage, int damage, int effect, int *out2) {
  Ch::Character c;
  c.set_Hp(hp), c.set_mindamage(mindamage);
  I::Weapon w;
  w.ready(damage, effect, 1);
  I::Bullet b;
  b.shot({0, 0, 0}, 1, w, 0);
  c.hit(b);
  out2[0] = c.get_Hp(), out2[1] = c.get_mindamage();
}

// gen_zombie (Character.hpp:866-871) at cor, `hits` x Character::hit with (damage, effect), then Zombie::punch in
// direction index `way` (Character.hpp:838-844): out = {Hp, mindamage, super, bullet cor[0..2], way, damage, effect,
// range, owner}
void ref_zombie(int super_, const int *cor, int hits, int damage, int effect, int way, int *out11) {
  Ch::Zombie z;
  Ch::gen_zombie(z, super_ != 0, {cor[0], cor[1], cor[2]}, "z");
  I::Weapon w;
  w.ready(damage, effect, 1);
  I::Bullet h;
  h.shot({0, 0, 0}, 1, w, 0);
  for (int i = 0; i < hits; ++i) z.hit(h);
  I::Bullet b;
  z.punch(b, way);
  const std::vector<int> bc = b.get_cor();
  out11[0] = z.get_Hp(), out11[1] = z.get_mindamage(), out11[2] = z.is_super() ? 1 : 0;
  out11[3] = bc[0], out11[4] = bc[1], out11[5] = bc[2], out11[6] = b.get_way(), out11[7] = b.get_damage();
  out11[8] = b.get_effect(), out11[9] = b.get_range(), out11[10] = (int)b.get_owner();
}

}  // extern "C"
