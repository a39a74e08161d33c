/*
 * sf_agent_adapter.hpp — header-only C++ adapter: drive reference-style `class Agent` bots from the
 * batched GPU simulator through the C-ABI (strikeforce.h).
 *
 * It rebuilds, per (arena, agent), the contract the reference's compile-time plugin slot gives a bot
 * (all paths under StrikeForce-client/):
 *   gameplay::prepare(Human&)   bots/bot-0.5/Custom.hpp:161-165  -> set `action`, `new Agent()`
 *   gameplay::bot(Human&)       bots/bot-0.5/Custom.hpp:137-159  -> obs (32x31x31 floats) -> predict() -> action[idx]
 *   human_action()              gameplay.hpp:970-976,991-997     -> agent->update(act, manual)
 *   deleteAgent()               Character.hpp:333-338, gameplay.hpp:648-649,1477
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
    check(sf_observe(env_, obs_.data()), "sf_observe");
    for (size_t i = 0; i < agents_.size(); ++i) {
      act_[i] = 0;
      cmd_[i] = (uint8_t)'+';
      if (!agents_[i]) continue;
      if constexpr (detail::has_predict<AgentT>::value) {
        /* `obs` is owned by the caller and valid only during predict(), as in Custom.hpp:141-158 */
        one_.assign(obs_.begin() + (ptrdiff_t)(i * SF_OBS_FLOATS), obs_.begin() + (ptrdiff_t)((i + 1) * SF_OBS_FLOATS));
        int idx = agents_[i]->predict(one_);
        if (idx < 0 || idx >= (int)action_.size()) idx = 0;
        act_[i] = idx;
        cmd_[i] = (uint8_t)action_[(size_t)idx];
      }
    }
    check(sf_step(env_, cmd_.data()), "sf_step");
    /* human_action(): `act` = position of the command in `action`, 0 if absent (gameplay.hpp:971-975) */
    for (size_t i = 0; i < agents_.size(); ++i)
      if (agents_[i]) agents_[i]->update(act_[i], manual);
    check(sf_done(env_, done_.data()), "sf_done");
    int ended = 0;
    for (int a = 0; a < cfg_.arenas; ++a)
      if (done_[(size_t)a]) {
        ++ended;
        /* end of play(): deleteAgent(); with auto_reset the next episode starts with prepare() again */
        for (int g = 0; g < cfg_.n_agents; ++g) {
          auto &slot = agents_[(size_t)a * (size_t)cfg_.n_agents + (size_t)g];
          slot.reset(cfg_.auto_reset ? new AgentT() : nullptr);
        }
      }
    return ended;
  }

  sf_env *env() const { return env_; }
  const std::vector<uint8_t> &last_commands() const { return cmd_; }
  const std::vector<uint8_t> &done() const { return done_; }

 private:
  static void check(int rc, const char *what) {
    if (rc != SF_OK) throw std::runtime_error(std::string(what) + ": " + sf_last_error());
  }
  sf_config cfg_;
  std::string action_;
  sf_env *env_ = nullptr;
  std::vector<std::unique_ptr<AgentT>> agents_;
  std::vector<int> act_;
  std::vector<uint8_t> cmd_, done_;
  std::vector<float> obs_, one_;
};

}  // namespace sf
#endif
