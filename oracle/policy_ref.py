"""PyTorch f32 restatement of the reference's bot network.  TEST INFRASTRUCTURE ONLY (like everything in oracle/):
only tests/, __graft_entry__.smoke() and bench.py's checker legs may import it; the product path is the HIP library.

Follows StrikeForce-client/bots/bot-0.5/Modules.hpp:
  ResB        :30-52     GameCNN :54-73     Backbone :75-136     AgentModel :138-180
and Agent.hpp:200-214 for the action distribution.

Two forms of the same arithmetic:
  * ``AgentModel`` — a torch.nn.Module built from the same torch modules, with the same registered names, as the
    reference's libtorch model, evaluated with batch 1 and holding its own memory exactly like one reference Agent.
    Its ``state_dict()`` keys are the names ``strikeforce_amd.policy.parameter_shapes()`` lists.
  * ``forward_batched`` — the same forward for B independent agents at once (every reduction is per agent), used as
    the fp32 reference the HIP kernels are compared with.  tests/test_policy_ref.py checks it against ``AgentModel``.

Pin status: PINNED on the reference itself.  Modules.hpp:26-180 compiles unedited against the libtorch inside the torch
wheel (oracle/ref_modules.py -> oracle/_ref/libsf_refmodules.so); tests/test_ref_modules.py holds ``AgentModel`` to the
reference's AgentModel bit for bit (probabilities, value, both recurrent states, over recurrent steps with
update_actions and a reset_memory, four parameter sets) and ``forward_batched`` to it within 1e-6 relative;
tests/golden/policy_vectors.json holds the reference's own outputs, which the HIP path reproduces (-m gpu).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HIDDEN, ACTIONS, CHANNELS, GRID = 160, 9, 32, 31
LAYER_INDEX = 3  # Modules.hpp:28
POV_CELLS = ((-1, 0), (0, -1), (0, 0), (0, 1), (1, 0))  # Modules.hpp:116


def _norm_all(y):
    """y * y.numel() / (y.abs().sum() + 1e-8) for a single sample   Modules.hpp:43"""
    return y * y.numel() / (y.abs().sum().detach() + 1e-8)


class ResB(nn.Module):  # Modules.hpp:30-51
    def __init__(self, hidden, layers):
        super().__init__()
        self.num_layers = layers
        for i in range(layers):
            self.add_module("lin%d" % i, nn.Linear(hidden, hidden))

    def forward(self, X):
        y = X.clone()
        x = _norm_all(y)
        for i in range(self.num_layers):
            y = torch.relu(getattr(self, "lin%d" % i)(x)) + x
            x = _norm_all(y)
        return x


class GameCNN(nn.Module):  # Modules.hpp:54-72
    def __init__(self, channels, d_out, layers):
        super().__init__()
        self.layers = layers
        for i in range(layers):
            self.add_module("conv%d" % i, nn.Conv2d(d_out if i else channels, d_out, 3, stride=2, padding=0, bias=False))

    def forward(self, x):
        y = x.clone()
        for i in range(self.layers):
            y = getattr(self, "conv%d" % i)(y)
        return y


class Backbone(nn.Module):  # Modules.hpp:75-135
    def __init__(self):
        super().__init__()
        self.cnn = GameCNN(CHANNELS, HIDDEN, 4)
        self.gru0 = nn.GRU(HIDDEN, HIDDEN, num_layers=1)
        self.combined_processor = nn.Sequential(nn.Linear(2 * HIDDEN + ACTIONS, HIDDEN))
        self.gru1 = nn.GRU(HIDDEN, HIDDEN, num_layers=1)
        self.reset_memory()

    def reset_memory(self):
        self.action_input = torch.zeros(ACTIONS)
        self.action_input[0] += 1
        self.h_state = [torch.zeros(1, 1, HIDDEN), torch.zeros(1, 1, HIDDEN)]

    def update_actions(self, one_hot):
        self.action_input = one_hot.clone()

    def forward(self, x):
        feat = self.cnn(x)
        feat = feat * HIDDEN / (feat.abs().sum().detach() + 1e-8)
        out_seq, self.h_state[0] = self.gru0(feat.view(1, 1, -1), self.h_state[0])
        out_seq = out_seq.view(-1)
        out_seq = out_seq * HIDDEN / (out_seq.abs().sum().detach() + 1e-8)
        y = []
        for e in POV_CELLS:
            for j in range(CHANNELS):
                y.append(x[0][j][GRID // 2 + e[0]][GRID // 2 + e[1]].clone())
        pov = torch.cat([torch.stack(y).view(-1), self.action_input])
        combined = torch.cat([out_seq + feat.view(-1), pov * HIDDEN / (pov.abs().sum().detach() + 1e-8)])
        gated = self.combined_processor(combined)
        gated = gated * HIDDEN / (gated.abs().sum().detach() + 1e-8)
        out, self.h_state[1] = self.gru1(gated.view(1, 1, -1), self.h_state[1])
        out = out.view(-1)
        return out * HIDDEN / (out.abs().sum().detach() + 1e-8) + gated


class AgentModel(nn.Module):  # Modules.hpp:138-179
    def __init__(self):
        super().__init__()
        self.backbone = Backbone()
        self.value = nn.Sequential(ResB(HIDDEN, LAYER_INDEX), nn.Linear(HIDDEN, 1))
        self.policy = nn.Sequential(ResB(HIDDEN, LAYER_INDEX), nn.Linear(HIDDEN, ACTIONS))

    def reset_memory(self):
        self.backbone.reset_memory()

    def update_actions(self, one_hot):
        self.backbone.update_actions(one_hot)

    def forward(self, x):
        gated = self.backbone(x)
        logits = self.policy(gated).view(-1)
        p = torch.softmax(logits, -1) + 1e-8
        v = torch.sigmoid(self.value(gated)).view(-1)
        return p, v


def model_from_parameters(params):
    m = AgentModel()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in params.items()}, strict=True)
    return m.eval()


def _rows_norm(x):
    return x * HIDDEN / (x.abs().sum(dim=1, keepdim=True) + 1e-8)


def _gru(x, h, w_ih, w_hh, b_ih, b_hh):
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    i_r, i_z, i_n = gi.chunk(3, 1)
    h_r, h_z, h_n = gh.chunk(3, 1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1 - z) * n + z * h


@torch.no_grad()
def forward_batched(params, obs, h, action_input):
    """obs [B,32,31,31], h [2,B,160], action_input [B,9] (numpy or tensors) -> probs [B,9], value [B], new h [2,B,160]."""
    P = {k: torch.as_tensor(np.asarray(v), dtype=torch.float32) for k, v in params.items()}
    x = torch.as_tensor(obs, dtype=torch.float32)
    h = torch.as_tensor(h, dtype=torch.float32)
    a = torch.as_tensor(action_input, dtype=torch.float32)
    B = x.shape[0]
    y = x
    for i in range(4):
        y = F.conv2d(y, P["backbone.cnn.conv%d.weight" % i], stride=2)
    feat = _rows_norm(y.reshape(B, HIDDEN))
    g = "backbone.gru0."
    h0 = _gru(feat, h[0], P[g + "weight_ih_l0"], P[g + "weight_hh_l0"], P[g + "bias_ih_l0"], P[g + "bias_hh_l0"])
    out_seq = _rows_norm(h0)
    c = GRID // 2
    pov = torch.cat([x[:, :, c + dy, c + dx] for dy, dx in POV_CELLS] + [a], dim=1)
    combined = torch.cat([out_seq + feat, pov * HIDDEN / (pov.abs().sum(dim=1, keepdim=True) + 1e-8)], dim=1)
    gated = _rows_norm(F.linear(combined, P["backbone.combined_processor.0.weight"], P["backbone.combined_processor.0.bias"]))
    g = "backbone.gru1."
    h1 = _gru(gated, h[1], P[g + "weight_ih_l0"], P[g + "weight_hh_l0"], P[g + "bias_ih_l0"], P[g + "bias_hh_l0"])
    out = _rows_norm(h1) + gated

    def head(name):
        xx = _rows_norm(out)
        for i in range(LAYER_INDEX):
            yy = torch.relu(F.linear(xx, P["%s.0.lin%d.weight" % (name, i)], P["%s.0.lin%d.bias" % (name, i)])) + xx
            xx = _rows_norm(yy)
        return F.linear(xx, P["%s.1.weight" % name], P["%s.1.bias" % name])

    probs = torch.softmax(head("policy"), dim=1) + 1e-8
    value = torch.sigmoid(head("value")).reshape(B)
    return probs.numpy(), value.numpy(), torch.stack([h0, h1]).numpy()


def action_weights(probs):
    """The weights Agent::predict hands to std::discrete_distribution   Agent.hpp:204-211"""
    v = np.array(probs, dtype=np.float32, copy=True)
    sc = np.float32(0.5) / (np.float32(1) - v[..., 0] + np.float32(1e-5))
    v[..., 1:] *= sc[..., None]
    v[..., 0] = 0.5
    return v
