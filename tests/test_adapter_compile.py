"""include/sf_agent_adapter.hpp compiles against reference-style Agent classes: the bot-0 form without predict()
(bots/bot-0/Agent.hpp:27-37) and the bot-0.5 form with predict/update/in_training/is_manual
(bots/bot-0.5/Agent.hpp:178,217,239,266)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <vector>
#include "sf_agent_adapter.hpp"
struct AgentNoPredict { void update(int, bool) {} bool in_training() { return false; } bool is_manual() { return false; } };
struct AgentPredict {
  int calls = 0;
  int predict(const std::vector<float> &obs) { ++calls; return obs.size() == SF_OBS_FLOATS ? 1 : 0; }
  void update(int, bool) {}
  bool in_training() { return false; }
  bool is_manual() { return false; }
};
template class sf::AgentRunner<AgentNoPredict>;
template class sf::AgentRunner<AgentPredict>;
static_assert(!sf::detail::has_predict<AgentNoPredict>::value, "bot-0 form");
static_assert(sf::detail::has_predict<AgentPredict>::value, "bot-0.5 form");
int main() { return 0; }
"""


def test_adapter_header_compiles(tmp_path):
    src = tmp_path / "adapter_check.cpp"
    src.write_text(SRC)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), str(src)])
