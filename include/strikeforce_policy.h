/* strikeforce_policy.h — C-ABI of the batched on-device policy network.
 *
 * The reference evaluates its bot network once per agent per tick on the host with libtorch, batch 1
 * (StrikeForce-client/bots/bot-0.5/Agent.hpp:178-214 predict(), Modules.hpp:54-179 AgentModel).  This
 * library evaluates the same network for every agent of an arena batch in one pass on the GPU, reading
 * the observation buffer sf_observe_device() wrote and producing the command chars sf_step_device()
 * consumes, so the 123 KB/agent observation never leaves HBM.  Inputs, outputs and state are f32.
 *
 * The convolution stack.  GameCNN::forward (Modules.hpp:66-71) applies four bias-free 3x3 / stride-2 convolutions with
 * nothing between them: one linear map of the 32 x 31 x 31 observation onto 160 features.  sf_policy_create composes the
 * four weight tensors into that map's matrix once (f64 on the device, rounded to f32), and a forward pass adds up one
 * row of it per non-zero float of the observation (about 1 % of the floats are non-zero): f32 fmaf chains per pair of
 * channels, combined in f64.  Same function as the layer-by-layer evaluation, 0.1 instead of 48.6 MFLOP per agent; the
 * difference from the reference's f32 layer-by-layer result is of the size of that result's own rounding (tests: |err| <=
 * 1e-6 + 5e-5 |ref| on probabilities, value and recurrent state against the reference's own AgentModel).
 * SF_POLICY_LAYERED=1 in the environment at sf_policy_create evaluates the four layers one after the other instead
 * (conv0 on the non-zeros; conv1 and conv2 — at 16 384 rows and more — on the bf16 MFMA with every f32 operand split
 * into three bf16 parts, |err| <= 2e-6 * sum|a*w|; conv3 on the f32 MFMA; SF_POLICY_F32_CONV=1 keeps all of them on the
 * f32 pipe): the cross-check path, held to the same tests.
 * Everything behind the convolutions (two GRU cells, combined_processor, the ResB heads) runs on the f32-input MFMA
 * (bit for bit an fmaf chain).  Every agent keeps its own recurrent state exactly as one
 * reference `Agent` object does.  Inference only: the PPO learner (Agent.hpp:270-420) is out of scope.
 *
 * Same conventions as strikeforce.h: plain pointers and sizes, 0 = success, sf_last_error() for text,
 * no CPU path (SF_ERR_DEVICE without a GPU).
 */
#ifndef STRIKEFORCE_POLICY_H
#define STRIKEFORCE_POLICY_H

#include <stdint.h>

#include "strikeforce.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SF_POLICY_ABI_VERSION 1
#define SF_POLICY_HIDDEN 160   /* hidden_size   Modules.hpp:83,144 */
#define SF_POLICY_ACTIONS 9    /* num_actions   Modules.hpp:83,144; action string "+xzqeawsd" Custom.hpp:162 */
#define SF_POLICY_CONVS 4      /* GameCNN(num_channels, hidden_size, 4, grid_x)  Modules.hpp:86 */
#define SF_POLICY_RES_LAYERS 3 /* LAYER_INDEX   Modules.hpp:28 */
#define SF_POLICY_POV (5 * SF_OBS_CHANNELS + SF_POLICY_ACTIONS) /* 169: the 5 centre cells + last action  Modules.hpp:115-122 */

/* Host pointers to the parameters, in the layouts torch uses for the reference's modules (row-major,
 * f32).  Names follow the reference's register_module() names. */
typedef struct sf_policy_weights {
  int32_t abi_version;
  /* backbone.cnn.conv{i}.weight: conv0 [160][32][3][3], conv1..3 [160][160][3][3]; stride 2, no padding, no
   * bias (Modules.hpp:58-64) */
  const float *conv_w[SF_POLICY_CONVS];
  /* backbone.gru{g}.weight_ih_l0 / weight_hh_l0 [480][160], bias_ih_l0 / bias_hh_l0 [480]; gate order r,z,n
   * (torch::nn::GRU, Modules.hpp:87,91) */
  const float *gru_w_ih[2], *gru_w_hh[2], *gru_b_ih[2], *gru_b_hh[2];
  /* backbone.combined_processor.0.weight [160][2*160+9], .bias [160]  (Modules.hpp:88-90) */
  const float *comb_w, *comb_b;
  /* value.0.lin{i}.weight [160][160], .bias [160]; value.1.weight [1][160], .bias [1]   (Modules.hpp:147-149) */
  const float *value_res_w[SF_POLICY_RES_LAYERS], *value_res_b[SF_POLICY_RES_LAYERS], *value_w, *value_b;
  /* policy.0.lin{i}.weight, .bias; policy.1.weight [9][160], .bias [9]                 (Modules.hpp:150-152) */
  const float *policy_res_w[SF_POLICY_RES_LAYERS], *policy_res_b[SF_POLICY_RES_LAYERS], *policy_w, *policy_b;
} sf_policy_weights;

typedef struct sf_policy sf_policy;

/* new AgentModel + load of its parameters (Agent.hpp:77-110), for `max_agents` independent agents (each has
 * its own h_state[2] and action_input, Modules.hpp:81).  Memory starts as after reset_memory(). */
int sf_policy_create(const sf_policy_weights *w, int32_t max_agents, int32_t device, sf_policy **out);
void sf_policy_destroy(sf_policy *p);

/* Backbone::reset_memory() (Modules.hpp:95-100) for the agents whose byte in d_mask (device) is non-zero; NULL resets
 * every agent.  Call it with the done flags when an arena restarts: the reference builds a new Agent per game
 * (Custom.hpp prepare()).  d_mask must hold `max_agents` bytes (the count given to sf_policy_create): every agent of
 * the policy is looked at.  A caller that runs fewer agents, with a mask of that many bytes (the sf_done_device output
 * of a smaller env), uses sf_policy_reset_memory_n: only agents [0, agents) are looked at. */
int sf_policy_reset_memory(sf_policy *p, const uint8_t *d_mask);
int sf_policy_reset_memory_n(sf_policy *p, const uint8_t *d_mask, int32_t agents);

/* AgentModel::forward (Modules.hpp:106-134,169-178) for agents [0, agents): d_obs is the device buffer of
 * sf_observe_device ([agents][32][31][31] f32); writes d_probs [agents][9] (softmax + 1e-8) and d_value [agents]
 * (sigmoid), and advances every agent's h_state. */
int sf_policy_forward(sf_policy *p, const float *d_obs, int32_t agents, float *d_probs, float *d_value);

/* sf_policy_forward on the observation in list form (sf_observe_sparse_device: d_keys / d_vals [agents][cap],
 * d_counts [agents], d_pov [agents][160]): the convolution stack takes the non-zeros as they come instead of scanning a
 * dense 123 KB buffer per agent for them, and the five centre cells come from d_pov.  Same results as the dense call,
 * bit for bit (same values applied in the same order).  cap is at most SF_POLICY_LIST_MAX (SF_ERR_ARG otherwise): with
 * that, "the list did not fit" means the same here as in sf_observe_overflow_device — count > cap, or the 0xffffffff
 * marker — and the two libraries never disagree about which agents fall back to the dense row.  Such an agent is
 * evaluated on an empty observation and counted: sf_policy_sparse_overflows() returns (and
 * clears) that count — non-zero means the caller should have taken the dense pair of calls for that step. */
#define SF_POLICY_LIST_MAX 2048 /* an observation of the BASELINE configurations has ~250-300 non-zeros */
int sf_policy_forward_sparse(sf_policy *p, const uint32_t *d_keys, const float *d_vals, const uint32_t *d_counts,
                             const float *d_pov, int32_t cap, int32_t agents, float *d_probs, float *d_value);
/* The same with a dense fallback, so that no agent is ever evaluated on a blank window: the agents whose list did not
 * fit are redone from their rows of d_dense (what sf_observe_overflow_device wrote: [agents][32][31][31]; the other rows
 * are never read) by a second launch that is idle when every list fitted.  Results equal the dense pair of calls
 * bit for bit for every agent; nothing is counted in sf_policy_sparse_overflows. */
int sf_policy_forward_sparse_or_dense(sf_policy *p, const uint32_t *d_keys, const float *d_vals, const uint32_t *d_counts,
                                      const float *d_pov, int32_t cap, int32_t agents, const float *d_dense,
                                      float *d_probs, float *d_value);
int sf_policy_sparse_overflows(sf_policy *p, int32_t *count);

/* The tail of Agent::predict() + Agent::update() (Agent.hpp:200-222): v[0] = 0.5, v[i>0] *= 0.5/(1-v[0]+1e-5),
 * draw from discrete_distribution(v) (greedy != 0: arg-max of v instead), store the one-hot as the agent's
 * action_input (update_actions) and write command char action_string[a] to d_cmd[agent] (gameplay::bot,
 * Custom.hpp:160-163).  The reference seeds std::mt19937 from std::random_device per call, so only the
 * distribution is specified; here draw i of agent a is a counter-based hash of (seed, a, i).
 * d_action (device, int32 per agent) may be NULL. */
int sf_policy_act(sf_policy *p, const float *d_probs, int32_t agents, const char *action_string, uint64_t seed,
                  int32_t greedy, uint8_t *d_cmd, int32_t *d_action);

/* Agent::predict() + Agent::update() as the ONE call they are in the reference (Agent.hpp:200-222), for the closed loop
 * on the device: sf_policy_reset_memory + sf_policy_forward_sparse(_or_dense) + sf_policy_act in the forward's two
 * launches, with the same results bit for bit as the three calls in that order (tests/test_gpu_policy.py).
 *   d_keys .. cap   the observation lists (sf_observe_sparse_device), as sf_policy_forward_sparse takes them
 *   d_dense         optional: dense rows for the lists that did not fit (sf_observe_overflow_device); without it such
 *                   agents are evaluated on a blank window and counted (sf_policy_sparse_overflows)
 *   d_reset_mask    optional, one byte per agent (sf_done_device's layout): non-zero = this agent's game restarted, it
 *                   starts from a new Agent's memory (zero state, "no action" as its last action) — gameplay.hpp:481
 *   d_reset_words   optional, the same flags where the environment keeps them (sf_done_view_device): agent a reads
 *                   word (a / reset_group) * reset_stride — saves the sf_done_device launch
 *   action_string, seed, greedy, d_cmd, d_action (may be NULL)   as in sf_policy_act
 *   d_probs, d_value   as in sf_policy_forward_sparse
 * The masks are read by the launch, in stream order: what sf_step_device left there is what counts. */
typedef struct sf_policy_predict_io {
  const uint32_t *d_keys;
  const float *d_vals;
  const uint32_t *d_counts;
  const float *d_pov;
  int32_t cap;
  const float *d_dense;
  const uint8_t *d_reset_mask;
  const int32_t *d_reset_words;
  int32_t reset_stride, reset_group;
  const char *action_string;
  uint64_t seed;
  int32_t greedy;
  float *d_probs, *d_value;
  uint8_t *d_cmd;
  int32_t *d_action;
} sf_policy_predict_io;
int sf_policy_predict_sparse(sf_policy *p, const sf_policy_predict_io *io, int32_t agents);

/* Read/write one agent's recurrent state for tests: h [2][160] and the action one-hot [9] (host buffers). */
int sf_policy_get_memory(sf_policy *p, int32_t agent, float *h, float *action_input);
int sf_policy_set_memory(sf_policy *p, int32_t agent, const float *h, const float *action_input);

int sf_policy_set_stream(sf_policy *p, void *hip_stream);
int sf_policy_synchronize(sf_policy *p);

/* Device time of the matrix kernels of the forward passes since the last call (HIP events on the library's
 * stream): *ms = summed duration of the MFMA GEMM launches, *flop = their algorithmic flop count
 * (2*M*N*K each), *launches = how many. */
int sf_policy_kernel_time(sf_policy *p, int32_t enable, float *ms, double *flop, int32_t *launches);
/* The same, split by matrix pipe: index 0 = launches on the f32 MFMA (k_gemm), index 1 = launches of the bf16-split
 * kernel (k_gemm_b3: conv1 / conv2 at M >= 16 384).  flop[] is the algorithmic 2*M*N*K in both; the split kernel
 * executes six bf16 products per algorithmic one. */
int sf_policy_kernel_time_ex(sf_policy *p, int32_t enable, float ms[2], double flop[2], int32_t launches[2]);
/* The same by kernel: 0 k_gemm (f32 MFMA) + fix-ups, 1 k_gemm_b3 (conv1 / conv2 on the bf16 pipe; layered form only),
 * 2 the launch that takes the observation's non-zeros — k_feat_list, the composed convolution stack (layered form:
 * k_conv0_sparse); flop[2] = 0: the useful work depends on the lists, 2 * 160 per non-zero — 3 k_tail (GRU cells +
 * combined_processor + heads; conv3 too in the layered form). */
int sf_policy_kernel_time_by_kernel(sf_policy *p, int32_t enable, float ms[4], double flop[4], int32_t launches[4]);

/* The matrix kernel on its own, for unit tests and roofline measurements: C[M][ldc] = A[M][lda] * W[N][K]^T + bias
 * (bias may be NULL), all device pointers, f32; K % 32 == 0, N % 160 == 0, lda % 4 == 0.  Every Linear / GRU gate
 * product of the network goes through this path; the convolutions use the same kernel with an im2col gather. */
int sf_policy_gemm(sf_policy *p, const float *d_a, int32_t lda, const float *d_w, const float *d_bias, float *d_c,
                   int32_t ldc, int32_t m, int32_t n, int32_t k);

/* The same product through the bf16-split kernel that conv1 and conv2 use once M >= 16 384 (k_gemm_b3: every f32
 * operand as three bf16 parts, six v_mfma_f32_32x32x16_bf16 per block instead of eight f32 ones, f32-level error:
 * |err| <= 2e-6 * sum|a*w|).  W is split on the device by this call (the network's own weights are split once at
 * sf_policy_create); synchronous.  For unit tests and roofline measurements. */
int sf_policy_gemm_split(sf_policy *p, const float *d_a, int32_t lda, const float *d_w, const float *d_bias, float *d_c,
                         int32_t ldc, int32_t m, int32_t n, int32_t k);

/* Test hook: GameCNN::forward alone (Modules.hpp:66-71) — the 160 features behind the four convolutions, before any
 * normalisation — for agents [0, agents) from the dense observations d_obs, into d_feat [agents][160] (device).  The
 * composed matrix by default, the four layers one after the other under SF_POLICY_LAYERED=1.  No recurrent state is
 * touched. */
int sf_policy_features(sf_policy *p, const float *d_obs, int32_t agents, float *d_feat);

int sf_policy_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* STRIKEFORCE_POLICY_H */
