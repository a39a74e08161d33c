// make_torch_archive.cpp — TEST TOOL: writes a checkpoint with libtorch's C++ `torch::save(module, path)`, the call the
// reference's bots use for model.pt (bots/bot-0.5/Agent.hpp:124,159-161), so that policy.load_checkpoint is tested on
// what the C++ writer really produces.  The module tree is built from a list of "name d0 d1 ..." lines on stdin (the
// parameter names AgentModel registers, Modules.hpp:37,62,87-91,147-152); parameter k, element i holds
// float(((i * 2654435761 + k * 40503) mod 2^32 >> 16) / 65536 - 0.5), which the test recomputes.
#include <torch/torch.h>

#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

struct Node : torch::nn::Module {
  std::map<std::string, std::shared_ptr<Node>> kids;
  std::shared_ptr<Node> child(const std::string &name) {
    auto it = kids.find(name);
    if (it != kids.end()) return it->second;
    auto n = std::make_shared<Node>();
    register_module(name, n);
    kids[name] = n;
    return n;
  }
};

int main(int argc, char **argv) {
  if (argc != 2) return 2;
  auto root = std::make_shared<Node>();
  std::string line;
  uint32_t k = 0;
  while (std::getline(std::cin, line)) {
    std::istringstream in(line);
    std::string name;
    if (!(in >> name)) continue;
    std::vector<int64_t> shape;
    int64_t d;
    while (in >> d) shape.push_back(d);
    int64_t n = 1;
    for (auto x : shape) n *= x;
    std::vector<float> v((size_t)n);
    for (int64_t i = 0; i < n; ++i)
      v[(size_t)i] = (float)((double)(((uint32_t)i * 2654435761u + k * 40503u) >> 16) / 65536.0 - 0.5);
    torch::Tensor t = torch::from_blob(v.data(), shape, torch::kFloat32).clone();
    std::shared_ptr<Node> node = root;
    size_t pos = 0, dot;
    while ((dot = name.find('.', pos)) != std::string::npos) {
      node = node->child(name.substr(pos, dot - pos));
      pos = dot + 1;
    }
    node->register_parameter(name.substr(pos), t);
    ++k;
  }
  torch::save(root, argv[1]);
  return 0;
}
