// ref_modules_wrap.cpp — TEST INFRASTRUCTURE (ours): C entry points around the reference's own bot network.
//
// "slice_modules.inc" is StrikeForce-client/bots/bot-0.5/Modules.hpp:26-180, cut out by line number and left
// unedited by oracle/ref_modules.py (`#include <torch/torch.h>`, `#define LAYER_INDEX 3`, ResB, GameCNN, Backbone,
// AgentModel).  Line 25 of that file (`#include "../../basic.hpp"`, which reaches SFML) is outside the range and
// nothing in the range uses it.  Nothing below restates the network: it constructs the reference's AgentModel,
// copies named parameters in, and calls the reference's forward / update_actions / reset_memory.
#include "slice_modules.inc"

#include <cstring>
#include <string>
#include <vector>

namespace {
struct Handle {
    AgentModel model;
    std::vector<std::string> names;
    Handle() : model(AgentModel()) {
        model->eval();
        for (auto &p : model->named_parameters()) names.push_back(p.key());
    }
};
}  // namespace

extern "C" {

void *rm_create() {
    try { return new Handle(); } catch (...) { return nullptr; }
}

void rm_destroy(void *h) { delete static_cast<Handle *>(h); }

int rm_param_count(void *h) { return (int)static_cast<Handle *>(h)->names.size(); }

// name into buf (NUL-terminated), dims into shape[0..3], returns the number of dims (or -1)
int rm_param_info(void *h, int i, char *buf, int cap, long long *shape) {
    auto *H = static_cast<Handle *>(h);
    if (i < 0 || i >= (int)H->names.size()) return -1;
    std::strncpy(buf, H->names[i].c_str(), cap - 1);
    buf[cap - 1] = 0;
    auto t = H->model->named_parameters()[H->names[i]];
    for (int d = 0; d < t.dim() && d < 4; ++d) shape[d] = t.size(d);
    return (int)t.dim();
}

// copy n floats into the parameter called `name`; 0 ok, -1 unknown name, -2 wrong size
int rm_set_param(void *h, const char *name, const float *src, long long n) {
    auto *H = static_cast<Handle *>(h);
    torch::NoGradGuard g;
    auto params = H->model->named_parameters();
    auto *t = params.find(name);
    if (!t) return -1;
    if (t->numel() != n) return -2;
    t->copy_(torch::from_blob(const_cast<float *>(src), {n}, torch::kFloat32).view(t->sizes()));
    return 0;
}

// AgentModel::forward on one observation [1][32][31][31] (the view Agent::predict builds, Agent.hpp:197);
// probs[9], value[1], and the recurrent state after the call h[2][160].
int rm_forward(void *h, const float *obs, float *probs, float *value, float *hstate) {
    auto *H = static_cast<Handle *>(h);
    try {
        torch::NoGradGuard g;
        auto x = torch::from_blob(const_cast<float *>(obs), {1, 32, 31, 31}, torch::kFloat32).clone();
        auto out = H->model->forward(x);
        auto p = out[0].contiguous(), v = out[1].contiguous();
        if (p.numel() != 9 || v.numel() != 1) return -2;
        std::memcpy(probs, p.data_ptr<float>(), 9 * sizeof(float));
        std::memcpy(value, v.data_ptr<float>(), sizeof(float));
        for (int k = 0; k < 2; ++k) {
            auto s = H->model->backbone->h_state[k].contiguous().view({-1});
            if (s.numel() != 160) return -3;
            std::memcpy(hstate + 160 * k, s.data_ptr<float>(), 160 * sizeof(float));
        }
        return 0;
    } catch (...) { return -1; }
}

// what Agent::update does with the chosen action (Agent.hpp:220-222): one-hot, AgentModel::update_actions
void rm_update_actions(void *h, int action) {
    auto *H = static_cast<Handle *>(h);
    auto one_hot = torch::zeros({9});
    one_hot[action] += 1;
    H->model->update_actions(one_hot);
}

void rm_reset_memory(void *h) { static_cast<Handle *>(h)->model->reset_memory(); }

}  // extern "C"
