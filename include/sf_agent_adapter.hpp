/*
 * sf_agent_adapter.hpp — header-only C++ adapter: drive reference-style `class Agent` bots from the
 * batched GPU simulator through the C-ABI (strikeforce.h).
 *
 * It rebuilds, per (arena, agent), the contract the reference's compile-time plugin slot gives a bot
 * (all paths under StrikeForce-client/):
 *   gameplay::prepare(Human&)   bots/bot-0.5/Custom.hpp:161-165  -> set `action`, `new Agent()`
 *   gameplay::bot(Human&)       bots/bot-0.5/Custom.hpp:137-159  -> obs (32x31x31 floats) -> predict() -> action[idx]
 *   human_action()              gameplay.hpp:970-976,991-997     -> agent->update(act, manual)
 *   deleteAgent()               Character.hpp:333-338, gameplay.hpp:648-649,1477 -> a human that dies loses its Agent
 *                               (sf_agent_alive) and is never asked again (gameplay.hpp:985,991)
 * Call order of one iteration, as in the reference (gameplay.hpp:955-958,970-999):
 *   predict(ind) on the loop-top state; zombie_action ... first update_bull; update(ind); then for every other living
 *   agent in slot order predict(i) on THAT state, update(i); then the humans' sweep.
 * Agents other than `ind` in one process exist only in Squad mode built with USE_AGENT_IN_SQUAD_NPCS
 * (gameplay.hpp:1883-1885,1896-1898); there the adapter uses sf_step_begin / sf_step_end.  In Battle mode every agent
 * is the `ind` of its own client process and observes at its own loop top: one sf_observe, one sf_step.
 * The Agent type is the reference's own (e.g. bots/bot-0.5/Agent.hpp): `int predict(const std::vector<float>&)`,
 * `void update(int, bool)`, optional `bool in_training()`, `bool is_manual()`.  bot-0's Agent has no predict();
 * its Custom.hpp returns '+' (bots/bot-0/Custom.hpp:39-41) and so does this adapter.
 *
 * Usage:
 *     #include "bots/bot-0.5/Agent.hpp"          // unchanged reference bot
 *     #include "sf_agent_adapter.hpp"
 *     sf::AgentRunner<Agent> run(cfg, "+xzqeawsd");   // action string of Custom.hpp:162
 *     run.reset(tb, serial);
 *     for (;;) run.step();                            // observe -> predict -> sf_step -> update
 */
#ifndef SF_AGENT_ADAPTER_HPP
#define SF_AGENT_ADAPTER_HPP

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "strikeforce.h"

namespace sf {

namespace detail {
template <class A, class = void>
struct has_predict : std::false_type {};
template <class A>
struct has_predict<A, std::void_t<decltype(std::declval<A &>().predict(std::declval<const std::vector<float> &>()))>>
    : std::true_type {};
}  // namespace detail

template <class AgentT>
class AgentRunner {
 public:
  AgentRunner(const sf_config &cfg, std::string action) : cfg_(cfg), action_(std::move(action)) {
    if (action_.empty()) throw std::invalid_argument("empty action string");
    int rc = sf_create(&cfg_, &env_);
    if (rc != SF_OK) throw std::runtime_error(std::string("sf_create: ") + sf_last_error());
    const size_t n = (size_t)cfg_.arenas * (size_t)cfg_.n_agents;
    agents_.resize(n);
    act_.assign(n, 0);
    cmd_.assign(n, (uint8_t)'+');
    done_.assign((size_t)cfg_.arenas, 0);
    alive_.assign(n, 1);
    obs_.resize(n * SF_OBS_FLOATS);
    one_.resize(SF_OBS_FLOATS);
  }
  ~AgentRunner() {
    if (env_) sf_destroy(env_);
  }
  AgentRunner(const AgentRunner &) = delete;
  AgentRunner &operator=(const AgentRunner &) = delete;

  /* setup()+load_data(): every commanded human gets a fresh Agent — gameplay::prepare. */
  void reset(const uint64_t *tb, const uint64_t *serial) {
    check(sf_reset(env_, tb, serial), "sf_reset");
    for (auto &a : agents_) a.reset(new AgentT());
  }

  /* One iteration of play()'s loop for all arenas.  Returns the number of arenas whose episode ended. */
  int step(bool manual = false) {
    const size_t na = (size_t)cfg_.n_agents;
    const bool mid = cfg_.mode == SF_MODE_SQUAD && cfg_.n_agents > 1;  // agents other than `ind` are asked inside human_action
    check(sf_observe(env_, obs_.data()), "sf_observe");
    for (size_t i = 0; i < agents_.size(); ++i) {
      act_[i] = 0;
      cmd_[i] = (uint8_t)'+';
      if (!mid || i % na == (size_t)cfg_.ind) ask(i);  // bot(hum[ind]) at the loop top, gameplay.hpp:955-958
    }
    if (!mid) {
      check(sf_step(env_, cmd_.data()), "sf_step");
      /* human_action(): `act` = position of the command in `action`, 0 if absent (gameplay.hpp:971-975) */
      for (size_t i = 0; i < agents_.size(); ++i)
        if (agents_[i]) agents_[i]->update(act_[i], manual);
    } else {
      check(sf_step_begin(env_), "sf_step_begin");  // zombie_action ... the first update_bull
      check(sf_agent_alive(env_, alive_.data()), "sf_agent_alive");
      check(sf_observe(env_, obs_.data()), "sf_observe");
      for (int a = 0; a < cfg_.arenas; ++a) {
        const size_t base = (size_t)a * na, me = base + (size_t)cfg_.ind;
        if (agents_[me]) agents_[me]->update(act_[me], manual);  // gameplay.hpp:970-976
        for (size_t g = 0; g < na; ++g) {                         // gameplay.hpp:985-999, slot order
          const size_t i = base + g;
          if (i == me) continue;
          if (!alive_[i]) agents_[i].reset();  // shot dead in the first half-tick: deleteAgent, gameplay.hpp:648-649
          if (!agents_[i]) continue;
          ask(i);
          agents_[i]->update(act_[i], false);
        }
      }
      check(sf_step_end(env_, cmd_.data()), "sf_step_end");
    }
    check(sf_done(env_, done_.data()), "sf_done");
    check(sf_agent_alive(env_, alive_.data()), "sf_agent_alive");
    int ended = 0;
    for (int a = 0; a < cfg_.arenas; ++a) {
      if (done_[(size_t)a]) {
        ++ended;
        /* end of play(): deleteAgent(); with auto_reset the next episode starts with prepare() again */
        for (size_t g = 0; g < na; ++g) agents_[(size_t)a * na + g].reset(cfg_.auto_reset ? new AgentT() : nullptr);
      } else {
        /* hit_human(): a human other than `ind` that died has lost its Agent for good (gameplay.hpp:648-649) */
        for (size_t g = 0; g < na; ++g)
          if (!alive_[(size_t)a * na + g] && g != (size_t)cfg_.ind) agents_[(size_t)a * na + g].reset();
      }
    }
    return ended;
  }

  sf_env *env() const { return env_; }
  const std::vector<uint8_t> &last_commands() const { return cmd_; }
  const std::vector<uint8_t> &done() const { return done_; }
  const std::vector<uint8_t> &alive() const { return alive_; }
  bool has_agent(int arena, int agent) const { return (bool)agents_[(size_t)arena * (size_t)cfg_.n_agents + (size_t)agent]; }

 private:
  /* gameplay::bot(): obs -> predict() -> action[idx]   bots/bot-0.5/Custom.hpp:137-159 */
  void ask(size_t i) {
    if (!agents_[i]) return;
    if constexpr (detail::has_predict<AgentT>::value) {
      /* `obs` is owned by the caller and valid only during predict(), as in Custom.hpp:141-158 */
      one_.assign(obs_.begin() + (ptrdiff_t)(i * SF_OBS_FLOATS), obs_.begin() + (ptrdiff_t)((i + 1) * SF_OBS_FLOATS));
      int idx = agents_[i]->predict(one_);
      if (idx < 0 || idx >= (int)action_.size()) idx = 0;
      act_[i] = idx;
      cmd_[i] = (uint8_t)action_[(size_t)idx];
    }
  }
  static void check(int rc, const char *what) {
    if (rc != SF_OK) throw std::runtime_error(std::string(what) + ": " + sf_last_error());
  }
  sf_config cfg_;
  std::string action_;
  sf_env *env_ = nullptr;
  std::vector<std::unique_ptr<AgentT>> agents_;
  std::vector<int> act_;
  std::vector<uint8_t> cmd_, done_, alive_;
  std::vector<float> obs_, one_;
};

}  // namespace sf
#endif
