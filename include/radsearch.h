/*
 * radsearch.h -- C ABI of the MI355X-native radiation-search PPO hot path (librs_hip.so).
 *
 * The reference (bentotten/radiation_ppo) has no plugin/FFI interface: its boundary is two Python
 * call surfaces.  This ABI sits directly behind them; every entry point names the reference
 * interface it replaces (paths relative to the reference root):
 *
 *   rs_step      <- RadSearch.step            gym_rad_search/gym_rad_search/envs/rad_search_env.py:443-728
 *   rs_reset     <- RadSearch.reset           rad_search_env.py:730-797 (+ create_obs :948-1011,
 *                                             sample_source_loc_pos :1013-1131)
 *   rs_set_epoch_end <- `env.epoch_end = True` algos/multiagent/train.py:482-484
 *   rs_gae       <- PPOBuffer.GAE_advantage_and_rewardsToGO   algos/multiagent/ppo.py:391-423
 *                   (discount_cumsum ppo.py:62-85), one call for the whole [T, N*A] buffer
 *
 * Conventions
 *   - plain C: pointers and sizes only, no torch types, no exceptions; return 0 = RS_OK.
 *   - every pointer marked "device" is HBM owned by the CALLER (PyTorch tensors in the Python host);
 *     the library borrows it for the duration of the call's stream-ordered work.
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises.
 *   - no allocation after rs_create; one handle per device; a handle is not thread-safe.
 *   - environments are independent: a multi-GPU job gives each rank its own handle with
 *     env_id_base = first global env id of the rank (results do not depend on the sharding).
 */
#ifndef RADSEARCH_H
#define RADSEARCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RS_ABI_VERSION 4
#define RS_OBS_DIM 11      /* [measurement, x/scale, y/scale, 8 range sensors]  rad_search_env.py:589-593 */
#define RS_NUM_ACTIONS 9   /* 0..7 directions, 8 idle                           rad_search_env.py:55-68  */
#define RS_MAX_AGENTS 8
#define RS_MAX_OBS 7       /* obstruction_count in {-1,0..7}                    rad_search_env.py:315,327 */

typedef void* rs_stream_t; /* hipStream_t */
typedef struct rs_handle rs_handle;

enum {
    RS_OK = 0,
    RS_ERR_INVALID_ARG = 1,
    RS_ERR_HIP = 2,
    RS_ERR_WORKSPACE = 3,
    RS_ERR_UNSUPPORTED = 4
};

/* per-env error bits accumulated on the device (conditions on which the reference raises) */
enum {
    RS_ENVERR_ZERO_DIST = 1,   /* detector on the source: intensity / 0 (rad_search_env.py:501)            */
    RS_ENVERR_IDLE_STALL = 2,  /* idle action stalled without collision (ValueError, :544-547, :562-565)  */
    RS_ENVERR_CORRECT_CAP = 4, /* correct_coords loop (:1278) hit the iteration cap                       */
    RS_ENVERR_BAD_ACTION = 8,  /* action outside 0..8 / -1 (assert, :616-627)                             */
    RS_ENVERR_NO_PATH = 16     /* no obstacle-free path detector -> source (cannot happen in a valid world) */
};

typedef struct rs_config {
    int32_t num_envs;                /* N: envs owned by this handle                                   */
    int32_t num_agents;              /* A: 1..RS_MAX_AGENTS (number_agents, rad_search_env.py:354)     */
    int32_t obstruction_count;       /* -1 => U{1..5} per epoch, 0 none, 1..7 (rad_search_env.py:327)  */
    int32_t enforce_grid_boundaries; /* rad_search_env.py:328                                          */
    int32_t bbox[4];                 /* x0,y0,x1,y1 in cm (default 0,0,2700,2700; :320-324)            */
    int32_t observation_area[2];     /* (200,500) (:325)                                               */
    int32_t falloff;                 /* 0 = as written in the reference, I/r (:501); 1 = I/r^2         */
    int32_t geom_group_size;         /* envs sharing one obstacle layout; 1 = reference-faithful       */
    uint32_t seed;                   /* Philox key word 0                                              */
    uint32_t env_id_base;            /* global id of local env 0 (Philox key word 1 = base + n)        */
    int32_t coord_noise;             /* coord_noise (rad_search_env.py:365, :569-580): N(0, 5 cm) on the two coordinates of the
                                        OBSERVATION (not of the state), a fresh pair per agent-step (Philox stream 96 + agent) */
    int32_t debug_spawn;             /* DEBUG (:387-389, :782-785, :1043-1090): source (500, 500), detector (1000, 1000), no
                                        minimum-distance / line-of-sight resampling, intensity 1e6, background 0             */
} rs_config;

/* optional per-agent info outputs of rs_step / rs_reset (any pointer may be NULL); all device, [N,A] */
typedef struct rs_info {
    uint8_t* out_of_bounds;       /* info["out_of_bounds"]        rad_search_env.py:607-612 */
    int32_t* out_of_bounds_count; /* info["out_of_bounds_count"]                            */
    uint8_t* blocked;             /* info["blocked"] (sticky per episode)                   */
    uint8_t* collision;           /* Agent.collision (:268)                                 */
} rs_info;

/* ---- lifecycle ----------------------------------------------------------------------------- */
const char* rs_strerror(int code);
int rs_abi_version(void);

/* bytes of device workspace a handle needs for `cfg` (0 on invalid cfg) */
size_t rs_state_bytes(const rs_config* cfg);

/* workspace: device memory of >= rs_state_bytes(cfg) bytes, 256-byte aligned, owned by the caller
 * and kept alive until rs_destroy.  Zero-fills it on `stream`; every env starts with epoch_end set
 * (rad_search_env.py:421) and must be rs_reset before the first rs_step. */
int rs_create(const rs_config* cfg, void* workspace, size_t workspace_bytes, rs_stream_t stream, rs_handle** out);
void rs_destroy(rs_handle* h);

/* introspection for tests/adapters: device pointer + shape of one SoA state field.
 * names: "src_x","src_y","intensity","bkg","iter_count","episode","tstep","err","done","epoch_end"
 * ([N]); "num_obs" ([G]); "rect" ([28,G] int32: (obstacle*4 + {x0,y0,x1,y1}) major); "dsrc" ([28,N] f64);
 * "x","y","oob_count" ([A,N] int32); "sp","prev" ([A,N] f64); "aflags" ([A,N] u8: bit0 blocked,
 * bit1 intersect, bit2 out_of_bounds, bit3 collision).  elem: 1,4,8 bytes. */
int rs_state_field(rs_handle* h, const char* name, void** dev_ptr, int32_t* elem_bytes, int32_t* rows, int32_t* cols);

/* ---- environment ----------------------------------------------------------------------------- */
/* `env.epoch_end = True` for every env: the next rs_reset of an env resamples its obstacle layout. */
int rs_set_epoch_end(rs_handle* h, rs_stream_t stream);

/* Reset the envs whose mask byte is non-zero (mask == NULL: all).  For those envs writes the initial
 * observation (the reference's `step(None)`), reward, team reward, done and info rows; other rows are
 * left untouched.  obs [N,A,11] f32, reward [N,A] f32, team [N] f32, done [N,A] u8 (any may be NULL). */
int rs_reset(rs_handle* h, const uint8_t* mask, float* obs, float* reward, float* team, uint8_t* done,
             const rs_info* info, rs_stream_t stream);

/* RadSearch.refresh_environment (rad_search_env.py:799-874): start the masked envs' episodes from SAVED parameters
 * instead of sampling them (the evaluation harness replays its test-environment sets this way, evaluate.py:346).
 * src_xy, det_xy [N][2] int32 (cm), intensity, bkg [N] int32; num_obs [N] + rects [N][7][4] int32 (x0,y0,x1,y1) replace
 * the obstacle layout, or both NULL to keep the env's current one (the reference's num_obs = 0 default).  Outputs as
 * rs_reset (the observation of the idle step the reference takes); afterwards iter_count = 1 as in the reference. */
int rs_refresh(rs_handle* h, const uint8_t* mask, const int32_t* src_xy, const int32_t* det_xy, const int32_t* intensity,
               const int32_t* bkg, const int32_t* num_obs, const int32_t* rects, float* obs, float* reward, float* team,
               uint8_t* done, const rs_info* info, rs_stream_t stream);

/* One lock-step of all N envs.  actions [N,A] int8: 0..8 (-1 == idle 8) = the dict form of RadSearch.step with the
 * collision rule (rad_search_env.py:645-659); 9 = the reference's step(None) for that agent (no move, stale distances,
 * :528-567); 16 + a (a in 0..8) = the single-int form `step(a)` (:676-690): action a for the agent, and when ANY agent of
 * an env carries this form no collision rule is applied in that env (the reference calls agent_step without
 * proposed_coordinates there).  Outputs as rs_reset.
 * done[n,a] is the env-wide latch as seen when agent a returned (rad_search_env.py:509,613). */
int rs_step(rs_handle* h, const int8_t* actions, float* obs, float* reward, float* team, uint8_t* done,
            const rs_info* info, rs_stream_t stream);

/* Uniforms for action sampling, u[n,a] in [0,1) with 24 random bits, from the env's own Philox stream:
 * key (seed, global env id), counter (0, step index of the NEXT rs_step, episode, 32 + a).  The host
 * (and the fused collector) sample a = #{j : cdf_j <= u}, so a trajectory is a pure function of
 * (seed, env id, policy).  Replaces Categorical(probs).sample() (FF_core.py:101-104). */
int rs_action_uniforms(rs_handle* h, float* u, rs_stream_t stream);

/* host copy of the OR of all per-env error bits (synchronises `stream`; debugging/adapters only) */
int rs_error_flags(rs_handle* h, rs_stream_t stream, uint32_t* flags_out);

/* ---- PPO buffer math ------------------------------------------------------------------------ */
/* GAE(lambda) advantages and rewards-to-go for a whole rollout buffer, time-major [T, M] (M = N*A
 * columns, one trajectory stream per column).  cut[t,m] != 0: the trajectory of column m ends after
 * step t and last_val[t,m] is its bootstrap value (V(s_T) on timeout / epoch cut, 0 on terminal,
 * train.py:462-487).  Computed in float64 per column exactly as scipy.lfilter does, stored float32. */
int rs_gae(const float* rew, const float* val, const uint8_t* cut, const float* last_val, float* adv, float* ret,
           int32_t T, int32_t M, double gamma, double lam, rs_stream_t stream);

/* ---- policy (FF_core.ActorCritic, NeuralNetworkCores/FF_core.py:42-129) ---------------------- */
/* One 2x64 tanh MLP in torch layout: w1 [64,11], b1 [64], w2 [64,64], b2 [64], w3 [out,64], b3 [out]
 * (row-major [out][in] float32, i.e. nn.Linear.weight / .bias of actor.0/.2/.4 or critic.0/.2/.4). */
typedef struct rs_mlp_params {
    const float *w1, *b1, *w2, *b2, *w3, *b3;
} rs_mlp_params;

/* actor logits [M,8] (pre-softmax: actor.4 output) and critic value [M] for M observations x [M,11] on the
 * matrix cores (v_mfma_f32_32x32x2_f32, exact f32).  Replaces ActorCritic.actor / .critic forward
 * (FF_core.py:95-129).  Any of logits / value may be NULL. */
int rs_policy_forward(const rs_mlp_params* actor, const rs_mlp_params* critic, const float* x, int32_t M,
                      float* logits, float* value, rs_stream_t stream);

/* ---- fused on-device collector ------------------------------------------------------------------
 * One launch = one epoch of the reference's collector loop (algos/multiagent/train.py:332-548) for all N
 * envs of the handle (single agent, N % 64 == 0, geom_group_size == 1): per lock-step the MLP forward on the
 * matrix cores, Philox inverse-CDF action sampling, the env step, the per-episode Welford standardisation
 * (RADTEAM_core.py:188-277, train.py:334-341,432-436), episode/epoch cut rules (train.py:394-405), the
 * bootstrap value (train.py:462-487), the masked reset (train.py:530) and the buffer row (PPOBuffer.store,
 * ppo.py:339-381).  Nothing leaves HBM; the policy weights are read once per launch. */
typedef struct rs_rollout_args {
    int32_t steps_per_epoch;    /* T */
    int32_t steps_per_episode;  /* L */
    /* rollout buffer, time-major (device, caller-owned) */
    float* obs;        /* [T,N,11] standardised observation fed to the networks */
    int64_t* act;      /* [T,N] */
    float* logp;       /* [T,N] */
    float* val;        /* [T,N] */
    float* rew;        /* [T,N] */
    float* last_val;   /* [T,N] bootstrap value where cut != 0 */
    uint8_t* cut;      /* [T,N] 1 = trajectory ends after this step */
    float* source_tar; /* [T,N,2] */
    /* collector state carried from epoch to epoch (device, caller-owned, [N]) */
    float* cur_obs;    /* [N,11] raw observation of the current state (from rs_reset / previous launch) */
    double* w_count;   /* Welford count, mean, M2, std */
    double* w_mean;
    double* w_sq;
    double* w_std;
    int32_t* steps_in_ep;
    float* ep_ret;
    /* per-env epoch statistics written by the launch ([N]) */
    int32_t* done_count;
    int32_t* oob_count;
    double* ep_ret_sum;
    double* ep_len_sum;
    int32_t* ep_count;
    /* per-env statistics of the returns of the episodes that ended in this launch (logger columns StdEpRet / MaxEpRet /
     * MinEpRet, train.py:620): sum of squares, max (-inf when none ended), min (+inf when none ended) */
    double* ep_ret_sq_sum;
    float* ep_ret_max;
    float* ep_ret_min;
} rs_rollout_args;

int rs_rollout(rs_handle* h, const rs_mlp_params* actor, const rs_mlp_params* critic, const rs_rollout_args* args,
               rs_stream_t stream);

/* ---- fused PPO loss + gradients ------------------------------------------------------------------
 * One pass over the whole batch (M samples) computes the loss of AgentPPO.update_rada2c
 * (algos/multiagent/ppo.py:1206-1225) in its batched form
 *     L = -( sum_i w_i min(r_i A_i, clip(r_i, 1-c, 1+c) A_i)  -  vf * sum_i w_i (V_i - R_i)^2  +  alpha * sum_i w_i H_i )
 * (w_i = 1 / (#ranks * #episodes of the sample's rank * length of its episode): the reference's mean over
 * ranks of the mean over episodes of per-episode means; H_i enters as a DETACHED scalar exactly as in the
 * reference, `ent = pi.entropy().detach().mean().item()` ppo.py:1216, so alpha moves the loss value but
 * contributes no gradient), its statistics, and dL/dtheta for every parameter of
 * the actor and the critic: forward, loss, backward and the weight-gradient GEMMs (contraction over samples,
 * operands transposed through LDS) all on the matrix cores, gradients reduced deterministically.
 * Replaces loss_pi.backward() (ppo.py:1253-1254) for the FF_core networks.
 *   grads: float32 [10441] = actor {w1 704, b1 64, w2 4096, b2 64, w3 512, b3 8} then critic {704, 64, 4096, 64, 64, 1}
 *   stats: float64 [5] = {approx_kl, entropy, clip fraction, value loss, total loss}   (weighted sums)
 *   workspace: device scratch of rs_ppo_grad_workspace_bytes() bytes. */
typedef struct rs_ppo_batch {
    const float* x;        /* [M,11] */
    const int64_t* act;    /* [M] */
    const float* adv;      /* [M] normalised advantages */
    const float* ret;      /* [M] */
    const float* logp_old; /* [M] */
    const float* w;        /* [M] */
    int32_t M;
    float clip_ratio, alpha, vf_coef;
} rs_ppo_batch;

size_t rs_ppo_grad_workspace_bytes(void);
/* stop_flag (device int32, may be NULL): when non-zero at launch the call is a no-op (early stop already hit).
 * grads: float32 [RS_PPO_GRAD_FLOATS] = the 10441 gradients (actor w1 b1 w2 b2 w3 b3, critic ...) followed by the five
 * statistics once more as float32 (hi, lo) pairs (stats[q] = hi + lo) and zero padding: under data parallelism ONE
 * all-reduce (SUM) of this buffer is the reference's mpi_avg_grads + mpi_avg(kl) (ppo.py:1250-1256). */
#define RS_PPO_STATS_TAIL 16
#define RS_PPO_GRAD_FLOATS (10441 + RS_PPO_STATS_TAIL)
int rs_ppo_grad(const rs_mlp_params* actor, const rs_mlp_params* critic, const rs_ppo_batch* batch, float* grads,
                double* stats, void* workspace, const int32_t* stop_flag, rs_stream_t stream);

/* Device-side state of one update_agent call (algos/multiagent/ppo.py:789-796): lives in HBM so the whole
 * "<= train_pi_iters Adam steps with KL early stop" loop can be enqueued without a host round trip. */
typedef struct rs_update_state {
    int32_t adam_step;   /* optimiser step count (bias correction), persists across epochs            */
    int32_t stopped;     /* 1 once approx_kl >= threshold was seen in this update (reset by the host)  */
    int32_t iters;       /* loss evaluations done in this update == the reference's `kk`               */
    int32_t pad;
    double last_stats[5]; /* stats of the last evaluated iteration {kl, entropy, clipfrac, vloss, loss} */
} rs_update_state;

/* One iteration of the reference's early-stopped loop (ppo.py:1250-1261) after rs_ppo_grad:
 *   if state->stopped: nothing.  else iters += 1, last_stats = stats;
 *   if stats[0] (approx_kl, already reduced over ranks) < kl_threshold: Adam step (torch.optim.Adam
 *   semantics, betas (0.9, 0.999), eps 1e-8) on all 12 tensors with `grads`; else stopped = 1.
 * actor/critic point at the LIVE parameter tensors (updated in place); m, v: float32 [10441].
 * stats == NULL: take the statistics from the (hi, lo) pairs behind the gradients (the all-reduced bucket). */
int rs_adam_step(const rs_mlp_params* actor, const rs_mlp_params* critic, const float* grads, float* m, float* v,
                 const double* stats, rs_update_state* state, float lr, float kl_threshold, rs_stream_t stream);

/* ---- RAD-TEAM heat maps (MapsBuffer, NeuralNetworkCores/RADTEAM_core.py:395-932) ------------------------
 * Per env the maps every agent's MapsBuffer would hold are kept ONCE (all owners see the same observations, so
 * readings / visit counts / obstacles / combined-locations maps and the estimator state are identical across
 * owners); the per-owner location map is a one-hot at the owner's cell and others = combined - location.
 *   rs_maps_update  <- MapsBuffer.observation_to_map (:532-616) for every env (one call per select_action round)
 *   rs_maps_reset   <- MapsBuffer.reset (:510-523) for the masked envs
 *   rs_maps_stack   <- CNNBase.get_map_stack (:1791-1836): actor [N,A,6,X,Y] = {prediction, location, others,
 *                      readings, visits, obstacles}, critic [N,4,X,Y] = {combined, readings, visits, obstacles}
 * visit_table: float32 [(steps_per_episode+1)*A + 1], visit_table[c] = the reference's
 * normalize_incremental_logscale(current_value = 2c, base, 2) evaluated by the caller (host Python math.log). */
typedef struct rs_maps rs_maps;
size_t rs_maps_state_bytes(int32_t num_envs, int32_t num_agents, int32_t steps_per_episode, int32_t map_x, int32_t map_y);
int rs_maps_create(int32_t num_envs, int32_t num_agents, int32_t steps_per_episode, int32_t map_x, int32_t map_y,
                   double resolution_accuracy, const float* visit_table, void* workspace, size_t workspace_bytes,
                   rs_stream_t stream, rs_maps** out);
void rs_maps_destroy(rs_maps* m);
int rs_maps_reset(rs_maps* m, const uint8_t* mask, rs_stream_t stream);
/* obs [N,A,11] float32 (the env's observation rows); pred [N,A,2] float32 scaled coordinates or NULL;
 * mask [N] or NULL (all).  Detector cells come from the env handle's integer coordinates. */
int rs_maps_update(rs_maps* m, rs_handle* env, const float* obs, const float* pred, const uint8_t* mask, rs_stream_t stream);
int rs_maps_stack(rs_maps* m, float* actor_stack, float* critic_stack, rs_stream_t stream);
/* introspection for tests: "pred_cell","cell" ([N,A] int32), "combined","readings","visits","obstacles" ([N,X*Y] f32) */
int rs_maps_field(rs_maps* m, const char* name, void** dev_ptr, int32_t* elem_bytes, int32_t* rows, int32_t* cols);

/* ---- RAD-TEAM CNN trunk (NeuralNetworkCores/RADTEAM_core.py:962-1023 Actor, :1211-1271 Critic) ----------------
 * conv3x3(Cin->8, pad 1) - ReLU - maxpool 2x2/2 - conv3x3(8->16, pad 1) - ReLU - Flatten on 27x27 maps, forward and
 * weight-gradient backward, reading the resident shared maps directly (replaces actor[0:6] / critic[0:6] of the
 * nn.Sequential and their autograd for a whole batch; the three Linear layers after it stay library GEMMs).
 *   maps   [S][4][729] float32: combined, readings, visits, obstacles of each sample
 *   agent >= 0: actor input (6 channels, CNNBase.get_map_stack :1791-1836) = {one-hot(pcells[s][agent]) or empty when
 *               -1, one-hot(cells[s][agent]), combined - location, readings, visits, obstacles}; cells/pcells [S][A] int64
 *   agent  < 0: critic input (4 channels) = maps as they are; cells/pcells ignored
 *   w1 [8][Cin][3][3], b1 [8], w2 [16][8][3][3], b2 [16]: torch Conv2d layouts
 *   a2 [S][2704] (= Flatten of the ReLU'd conv2 output).  For training (all three non-NULL, or all NULL) the forward
 *   also writes what backward consumes: p1 [S][169][8] float32 pooled activations, amax [S][169][8] uint8 the winning
 *   pixel (0..3, row-major, first maximum) of each 2x2 pool window, relu_mask [S][169] uint16 bit c = (a2 channel c > 0).
 * Backward: da2 [S][2704] = dL/d(a2); slab [rs_cnn_trunk_slab_rows(S, Cin)][rs_cnn_trunk_slab_row(Cin)] receives
 * per-workgroup partial sums laid out {dW1 8*Cin*9 | db1 8 | dW2 1152 | db2 16}; the caller sums the rows.
 * wscratch: device scratch of rs_cnn_trunk_scratch_floats(Cin) floats (the weights re-laid-out for scalar loads). */
int32_t rs_cnn_trunk_slab_row(int32_t in_channels);
int32_t rs_cnn_trunk_slab_rows(int64_t num_samples, int32_t in_channels);
int32_t rs_cnn_trunk_scratch_floats(int32_t in_channels);
int rs_cnn_trunk_forward(const float* maps, const int64_t* cells, const int64_t* pcells, int32_t num_agents, int32_t agent,
                         int64_t num_samples, const float* w1, const float* b1, const float* w2, const float* b2, float* a2,
                         float* p1, uint8_t* amax, uint16_t* relu_mask, float* wscratch, rs_stream_t stream);
int rs_cnn_trunk_backward(const float* maps, const int64_t* cells, const int64_t* pcells, int32_t num_agents, int32_t agent,
                          int64_t num_samples, const float* w2, const float* da2, const uint16_t* relu_mask, const float* p1,
                          const uint8_t* amax, float* slab, float* wscratch, rs_stream_t stream);

/* ---- PFGRU location predictor -------------------------------------------------------------
 * One forward step of `self.model(obs_tensor, hidden)` in CNNBase.select_action (algos/test_cnn/RADTEAM_core.py:1872-1879):
 * PFGRUCell.forward (:1586-1631) = GRU-style particle update with reparameterised noise (:1517-1530), observation
 * likelihood + log-softmax (:1633-1641), soft resampling (:1466-1515), weighted particle mean -> hid_obs MLP (:1574-1584),
 * for every (owner, env) at once: 40 particles x 24 hidden units.
 *   weights   [A][RS_PFGRU_WEIGHT_FLOATS]  per-owner packed parameters (layout: csrc/rs_pfgru.hip; packer: pfgru.py)
 *   obs       [N][A][11]   the owner's own row supplies (reading, x, y)
 *   h, p      [A][N][6][40][4] particles, QUAD-major (a set's 40 particles x 24 units as six [40] x float4 slabs: unit u of particle q at
 *             ((u / 4) * 40 + q) * 4 + u % 4 -- coalesced for one particle per lane), [A][N][40] log weights; read; written back when
 *             carry_hidden != 0 (mask[n] != 0)
 *   base_key  [A][N], episode [N], calls [N]   counters of the draw hash (documented RNG deviation: the reference draws
 *             from torch's global generator)
 *   pred      [N][A][2]    location prediction (scaled coordinates, >= 0: the reference's MLP ends in a ReLU); with a mask only the
 *             masked envs' rows are written (the bootstrap round of the collectors, train.py:462-480: the others are not evaluated) */
#define RS_PFGRU_PARTICLES 40
#define RS_PFGRU_HIDDEN 24
#define RS_PFGRU_WEIGHT_FLOATS 3472
int rs_pfgru_step(const float* weights, const float* obs, float* h, float* p, const int64_t* base_key, const int64_t* episode,
                  const int64_t* calls, const uint8_t* mask, int32_t carry_hidden, double alpha, float* pred, int32_t num_envs,
                  int32_t num_agents, rs_stream_t stream);
/* A whole no-grad pass over an episode-major batch in ONE call: grad_step's PFGRU loop over an episode (RADA2C_core.py:555-558) for E
 * episodes of one predictor.  Enqueues rs_pfgru_reset and then the time steps t < steps on `stream`, four per launch (carried particle
 * sets staying in registers inside a launch; the step counter of step t = calls[t][.]; results identical to one rs_pfgru_step per t), each
 * launch over the first alive[t] episodes of its first step (episodes sorted by descending length: the ones still running are a prefix;
 * an episode that ends inside a launch's group of steps reports only its own steps).  obs [steps][E][11], calls [steps][E], pred [steps][E][2], alive: HOST array [steps].  Exists so that the
 * ~120 launches of a pass cost one library call: the policy loop of update_rada2c issues 40 such passes per update. */
int rs_pfgru_pass(const float* weights, const float* obs, float* h, float* p, const int64_t* base_key, const int64_t* episode,
                  const int64_t* calls, double alpha, float* pred, const int32_t* alive, int32_t steps, int32_t episodes, rs_stream_t stream);
/* The same step with the draws supplied instead of hashed: eps [A][N][40][24] = the reparameterisation noise, idx [A][N][40] = the
 * resampling indices (what FloatTensor.normal_ / torch.multinomial returned in a recorded run of the reference, :1485-1530).  The
 * arithmetic is the product kernel's (one template, two instantiations); used to hold it to tests/golden/pfgru.npz directly.  h is
 * quad-major as in rs_pfgru_step; eps stays particle-major. */
int rs_pfgru_step_recorded(const float* weights, const float* obs, float* h, float* p, const float* eps, const int32_t* idx,
                           const uint8_t* mask, int32_t carry_hidden, double alpha, float* pred, int32_t num_envs, int32_t num_agents,
                           rs_stream_t stream);
/* reset_hidden (RADTEAM_core.py:2030-2033, PFGRUCell.init_hidden :1643-1652) for the envs with mask[n] != 0 (all when null):
 * h0 ~ U[0,1) from the draw hash (written in rs_pfgru_step's quad-major layout), p0 = log(1/40).  episode[] / calls[] must already count
 * the new episode. */
int rs_pfgru_reset(float* h, float* p, const int64_t* base_key, const int64_t* episode, const int64_t* calls, const uint8_t* mask,
                   int32_t num_envs, int32_t num_agents, rs_stream_t stream);

/* All draws of one PFGRU training pass over an episode-major batch (the counter hash of rada2c.HashDraws: keys [E] int64):
 * h0 [E][40][24] initial particles (uniforms), eps [L][E][40][24] reparameterisation noise (standard normals), u [L][E][40]
 * resampling uniforms (float64).  Replaces torch.rand / FloatTensor.normal_ / torch.multinomial's generator in update_model
 * (algos/multiagent/ppo.py:1062-1079 via RADA2C_core.py:199-211,279-288) -- the documented RNG deviation. */
int rs_pfgru_draws(const int64_t* keys, int32_t episodes, int32_t steps, float* h0, float* eps, double* u, rs_stream_t stream);

/* One pass of AgentPPO.update_model's loss and its gradients (algos/multiagent/ppo.py:1062-1128 + loss.backward()): the PFGRU
 * unrolled through every episode of an episode-major batch, the loss
 *     total_e = l2w * sum_t bp_t |loc_t - tar_t|^2 + l1w * 10 * sum_t bp_t |loc_t - tar_t|_1 / n_e
 *             + elbo * (l2w * sum_{t,c} -log mean_p exp(-(part_tpc - tar_tc)^2 bp_t) + l1w * 10 * sum_{t,c} -log mean_p exp(-|.| bp_t)) / n_e
 * (n_e = 2 x the episode's length; loc = hid_obs(weighted particle mean), part = hid_obs of every resampled particle), and
 * d (w_ep[e] * total_e) / d parameters by back-propagation through time, resampling indices held constant as in autograd.
 *   weights [RS_PFGRU_TRAIN_WEIGHT_FLOATS]  packed parameters (layout: csrc/rs_pfgru_train.hip; packer: rada2c.py)
 *   obs [L][E][11] (columns 0..2 used), target [L][E][2], bp [L][E], lens [E] (1..L; steps beyond are never touched), w_ep [E]
 *   h0 / eps / u  the draws of rs_pfgru_draws; u may be NULL: idx[] is then INPUT -- the resampling indices to take (the ones
 *                 torch.multinomial returned in a recorded run of the reference, tests/golden/rada2c_core.npz)
 *   hs [L][E][6][40][4], ps [L][E][40], gates [L][E][24][40][4]  scratch (the resampled particle sets; the forward walk's gates z | r | n |
 *                 eps * softplus'(var), reloaded by the backward walk), idx [L][E][40] the resampling indices taken (output when u is given)
 *   loss [E] = w_ep[e] * total_e;  grads [E][RS_PFGRU_TRAIN_GRAD_FLOATS] = the episode's gradient slab:
 *   d[fc_z | fc_r] [48][28] (column 27 = bias) | d fc_n [48][28] | d hid_obs[0] [24][25] | d hid_obs[2] [2][25] | d fc_obs [28];
 *   the caller sums the slabs over the episodes. */
#define RS_PFGRU_TRAIN_WEIGHT_FLOATS 3680
#define RS_PFGRU_TRAIN_GRAD_FLOATS 3376
int rs_pfgru_train(const float* weights, const float* obs, const float* target, const float* bp, const int64_t* lens, const float* w_ep,
                   const float* h0, const float* eps, const double* u, float* hs, float* ps, float* gates, int32_t* idx, float* loss,
                   float* grads, int32_t steps, int32_t episodes, double alpha, double l2_weight, double l1_weight, double elbo_weight,
                   rs_stream_t stream);
/* The same pass with the draws of rs_pfgru_draws(keys, ...) evaluated INSIDE the forward walk instead of read from h0 / eps / u: same keys,
 * same counter hash, same arithmetic -- bit-identical loss, gradients and indices -- without the 104 bytes per particle-step the draws
 * launch writes and the walk reads back (8.2 GB per pass at 16 500 episodes; ABI version 4).  h0 [E][40][24]: scratch (the forward walk
 * leaves the initial particles there for the backward walk's last step). */
int rs_pfgru_train_keyed(const float* weights, const float* obs, const float* target, const float* bp, const int64_t* lens, const float* w_ep,
                         const int64_t* keys, float* h0, float* hs, float* ps, float* gates, int32_t* idx, float* loss, float* grads, int32_t steps,
                         int32_t episodes, double alpha, double l2_weight, double l1_weight, double elbo_weight, rs_stream_t stream);

/* ---- RAD-A2C GRU recurrence -----------------------------------------------------------------
 * The time loop of torch.nn.GRU(13, 24, 1) as SeqPt.forward / grad_step run it over whole episodes
 * (NeuralNetworkCores/RADA2C_core.py:377-381, :550-566) and of its back-propagation through time, for an episode-major batch
 * (one episode per lane).  Gate order r, z, n; gh = W_hh h + b_hh; n = tanh(gi_n + r * gh_n); h' = (1 - z) n + z h.
 *   rs_gru_forward : gi [L][E][72] (= X W_ih^T + b_ih, computed by the caller), h0 [E][24],
 *                    whh_t [24][80] (k-major W_hh^T, columns 72..79 zero), bhh [80] (b_hh, zero padded)
 *                    -> hs [L][E][24] (h_t), gates [L][E][96] = r | z | n | (W_hn h_{t-1} + b_hn)
 *   rs_gru_backward: dhs [L][E][24] = dL/dh_t from everything downstream, hs, gates, h0, whh [72][32] (W_hh, columns 24..31 zero)
 *                    -> dgi [L][E][72] = dL/d(gi), dgh [L][E][72] = dL/d(gh); the caller forms dW_ih = sum dgi^T x,
 *                       db_ih = sum dgi, dW_hh = sum dgh^T h_{t-1}, db_hh = sum dgh. */
#define RS_GRU_HIDDEN 24
int rs_gru_forward(const float* gi, const float* h0, const float* whh_t, const float* bhh, float* hs, float* gates, int32_t steps,
                   int32_t episodes, rs_stream_t stream);
int rs_gru_backward(const float* dhs, const float* hs, const float* gates, const float* h0, const float* whh, float* dgi, float* dgh,
                    int32_t steps, int32_t episodes, rs_stream_t stream);

/* One step of the RAD-A2C actor-critic behind the PFGRU, for every env (RNNModelActorCritic.step after the location prediction,
 * NeuralNetworkCores/RADA2C_core.py:528-548): h' = GRU(cat(x, loc), h) (SeqPt.forward :377-381), logits = Woms(h') (:363-366),
 * value = Valms(h') (:367-368), action by inverse CDF of softmax(logits) on the caller's uniform u (Categorical.sample :541-544 --
 * documented RNG deviation), logp = log_softmax(logits)[action].
 *   weights [RS_RNN_POLICY_WEIGHT_FLOATS]  packed parameters (layout: csrc/rs_rnn_policy.hip; packer: rada2c.py)
 *   x [N][11], loc [N][2], h [N][24];  u [N] (needed when act / logp are wanted)
 *   outputs, any may be NULL: h_out [N][24] (may alias h), logits [N][8], value [N], act [N] int64, logp [N] */
#define RS_RNN_POLICY_WEIGHT_FLOATS 5296
int rs_rnn_policy_step(const float* weights, const float* x, const float* loc, const float* h, const float* u, float* h_out,
                       float* logits, float* value, int64_t* act, float* logp, int32_t num_envs, rs_stream_t stream);
/* The same step reading agent a's rows straight out of the collectors' [N][A][.] tensors (x_stride / loc_stride / u_stride: floats
 * between consecutive envs' rows) and writing the action once more as int8 into rs_step's action row (act8 [N][act8_stride]);
 * mask [N] or NULL: only the envs with mask != 0 are evaluated (the bootstrap round of a lock-step: the others' outputs are untouched). */
int rs_rnn_policy_step_rows(const float* weights, const float* x, int32_t x_stride, const float* loc, int32_t loc_stride, const float* h,
                            const float* u, int32_t u_stride, float* h_out, float* value, int64_t* act, float* logp, int8_t* act8,
                            int32_t act8_stride, const uint8_t* mask, int32_t num_envs, rs_stream_t stream);

/* The heads of the RAD-A2C actor-critic, update_rada2c's per-sample loss (algos/multiagent/ppo.py:1191-1234: PPO-clip surrogate on
 * the policy head, vf_coef x squared error on the value head; the entropy term carries no gradient) and their back-propagation,
 * for `samples` (step, episode) rows behind the GRU:
 *   hs [S][24], act [S], adv [S], ret [S], logp_old [S], sample_weight [S] (w_ep / episode length; 0 on padded steps)
 *   -> dhs [S][24] = dL/dh (what rs_gru_backward takes), dfac [S][80] = d pre-tanh of the policy head [32] | of the value head [32]
 *      | d logits [8] | d value | zeros, tfac [S][64] = tanh outputs of the two heads: the caller forms the weight gradients
 *      (dW1 = dfac[:, :32]^T hs, dW2 = dfac[:, 64:72]^T tfac[:, :32], ...; biases = column sums of dfac);
 *      stats [ceil(S / 64)][8]: per-wave weighted sums of kl, entropy, clip fraction, value loss, surrogate, weight. */
int rs_a2c_heads_loss(const float* weights, const float* hs, const int64_t* act, const float* adv, const float* ret, const float* logp_old,
                      const float* sample_weight, float* dhs, float* dfac, float* tfac, float* stats, int64_t samples, double clip_ratio,
                      double vf_coef, rs_stream_t stream);

/* The forward trunk as the collectors use it: rs_cnn_trunk_prepare re-arranges a network's convolution weights into wscratch
 * (rs_cnn_trunk_scratch_floats floats) once per epoch, rs_cnn_trunk_infer is then ONE launch per select_action round (a2 only). */
int rs_cnn_trunk_prepare(int32_t in_channels, const float* w1, const float* b1, const float* w2, const float* b2, float* wscratch,
                         rs_stream_t stream);
int rs_cnn_trunk_infer(const float* maps, const int64_t* cells, const int64_t* pcells, int32_t num_agents, int32_t agent, int64_t num_samples,
                       const float* wscratch, float* a2, rs_stream_t stream);

/* The RAD-TEAM heads behind their first Linear layer (CNNBase.select_action, RADTEAM_core.py:1838-1892; Actor :1000-1023, Critic
 * :1250-1271): y1 [N][32] = Linear(2704, 32)(a2), pre-activation -> ReLU -> Linear(32, 16) (w2 [16][32], b2) -> ReLU -> Linear(16, out_dim)
 * (w3 [out_dim][16], b3).  out_dim = 8: the action logits -> log-softmax, inverse-CDF draw on u [n * u_stride] (a = #{j < 7: cdf_j <= u}),
 * act [N] int64, logp [N], act8 [n * act8_stride] (any NULL: not written).  out_dim = 1: the state value, written value_copies times at
 * value_stride floats.  mask [N] or NULL: only those envs. */
int rs_cnn_head(const float* y1, const float* w2, const float* b2, const float* w3, const float* b3, int32_t out_dim, const float* u,
                int32_t u_stride, int64_t* act, float* logp, int8_t* act8, int32_t act8_stride, float* value, int32_t value_copies,
                int64_t value_stride, const uint8_t* mask, int32_t num_envs, rs_stream_t stream);

/* The PPO-clip actor loss of the RAD-TEAM update behind the logits (AgentPPO.compute_loss_pi, algos/multiagent/ppo.py:966-1003) and its
 * derivative, one pass: logits [S][8], act [S], adv [S], logp_old [S], sample_weight [S] ->
 *   dlogits [S][8] = d (-sum_s w_s min(ratio_s adv_s, clip(ratio_s) adv_s)) / d logits,
 *   stats [ceil(S / 64)][4]: per-wave weighted sums of kl (logp_old - logp), entropy, clip fraction and the loss itself. */
int rs_actor_loss(const float* logits, const int64_t* act, const float* adv, const float* logp_old, const float* sample_weight,
                  float* dlogits, float* stats, int64_t samples, double clip_ratio, rs_stream_t stream);

/* _get_init_states (RADA2C_core.py:458-461) for the envs with mask[n] != 0 (all when NULL): GRU state h [A][N][24] ~
 * U(-scale, scale), scale = 1 / sqrt(24), from the counter hash of the env's key base_key [A][N] and its episode counter
 * episodes_begun [N] (documented RNG deviation: the reference draws from torch's global generator). */
int rs_gru_h0_reset(float* h, const int64_t* base_key, const int64_t* episodes_begun, const uint8_t* mask, double scale, int32_t num_envs,
                    int32_t num_agents, rs_stream_t stream);

/* ---- running observation statistics ------------------------------------------------------------
 * StatisticStandardization.update / standardize / reset (NeuralNetworkCores/RADTEAM_core.py:188-277) for num_envs x num_agents
 * independent streams, float64 state count / mean / sq (square_dist_mean) / std, one pass each:
 *   update      : streams of the envs with mask[n] != 0 (all when NULL) take the reading reading[i * stride], i = n * A + a
 *                 (Welford; the first sample sets the mean only; std = max(sqrt(sq / (count - 1)), 1))
 *   reset       : count = mean = sq = 0, std = 1 for the masked envs
 *   standardize : out[i * out_stride] = (float)((reading[i * stride] - mean[i]) / std[i]) */
int rs_welford_update(double* count, double* mean, double* sq, double* std, const float* reading, int64_t stride, const uint8_t* mask,
                      int32_t num_envs, int32_t num_agents, rs_stream_t stream);
int rs_welford_reset(double* count, double* mean, double* sq, double* std, const uint8_t* mask, int32_t num_envs, int32_t num_agents,
                     rs_stream_t stream);
int rs_welford_standardize(const double* mean, const double* std, const float* reading, int64_t stride, float* out, int64_t out_stride,
                           int32_t streams, rs_stream_t stream);

/* The logger statistics train() accumulates per epoch and agent id (algos/multiagent/train.py:386-398, :494-501, :519-526), one
 * lock-step of all envs: acc_oob[a] += sum_n out_of_bounds[n][a]; acc_done[a] += sum_n done[n][a]; over the envs whose episode
 * ended (over[n] != 0): ep_cnt += 1, ep_len += steps_in_ep[n], ret_sum[a] += r, ret_sq[a] += r^2, ret_max / ret_min (r =
 * ep_ret[n][a]).  float64 accumulators, fixed summation order. */
int rs_epoch_stats(const uint8_t* out_of_bounds, const uint8_t* done, const float* ep_ret, const int32_t* steps_in_ep, const uint8_t* over,
                   double* acc_oob, double* acc_done, double* ep_cnt, double* ep_len, double* ret_sum, double* ret_sq, double* ret_max,
                   double* ret_min, int32_t num_envs, int32_t num_agents, rs_stream_t stream);

/* PPOBuffer.store (algos/multiagent/ppo.py:296-389) for one lock-step of every (env, agent): row *t (a device-side counter) of the
 * time-major rollout buffers receives act [A][N], logp / value / bootstrap value (logp_val_boot [A][3][N]; the bootstrap value only
 * where boot[n] != 0, else 0), the observation x [N][A][11], the source location (src_x, src_y [N] int32 -> float), the reward
 * rew [N][A] and the cut flag cut [N]. */
int rs_store_rows(const int64_t* t, const int64_t* act, const float* logp_val_boot, const float* x, const int32_t* src_x, const int32_t* src_y,
                  const float* rew, const uint8_t* cut, const uint8_t* boot, int64_t* buf_act, float* buf_logp, float* buf_val,
                  float* buf_last_val, float* buf_obs, float* buf_source, float* buf_rew, uint8_t* buf_cut, int32_t num_envs,
                  int32_t num_agents, int32_t steps_per_epoch, rs_stream_t stream);

/* The element-wise bookkeeping of one collector lock-step between the library calls -- the epoch loop body of train_PPO.train
 * (algos/multiagent/train.py:332-548) for all envs at once: what the reference does in Python per env and step, as three launches.
 * Every pointer is a device buffer of the caller (addresses fixed for the life of the collector: the lock-step is replayed as a graph).
 *   rs_collect_pre        x <- obs with the reading (column 0) standardised by the running statistics (:334-341)
 *   rs_collect_post_step  after rs_step: ep_ret += reward (the team reward for every agent when team_reward != 0, :366-375),
 *                         steps_in_ep += 1, over = any agent's terminal flag | steps_in_ep == steps_per_episode (:387-405),
 *                         cut = over (everything at the epoch's last step), boot = timeouts (all cut envs at the epoch's last step:
 *                         the envs whose value is bootstrapped, :462-487); Welford update with the new readings (:432-436);
 *                         obs <- env_obs; xb <- its standardised form; pf_calls += 1 (the predictor bank's per-env call counter)
 *   rs_collect_post_reset after rs_reset(cut): pf_calls += boot (the bootstrap round's prediction); for the cut envs obs <- env_obs,
 *                         ep_ret = 0, steps_in_ep = 0, statistics restarted on the first reading (:504-548) and, with reset_hidden,
 *                         pf_episode += 1, pf_calls = 0, episodes_begun += 1 (the draw counters rs_pfgru_reset / rs_gru_h0_reset read);
 *                         t += 1 */
typedef struct {
    int32_t num_envs, num_agents, steps_per_episode, team_reward;
    const float* env_obs;         /* [N][A][11] the env's observation rows (output of rs_step / rs_reset) */
    const float* env_reward;      /* [N][A] */
    const float* env_team;        /* [N] */
    const uint8_t* env_done;      /* [N][A] */
    float* obs;                   /* [N][A][11] the collector's current observation */
    float* ep_ret;                /* [N][A] */
    int32_t* steps_in_ep;         /* [N] */
    double* w_count; double* w_mean; double* w_sq; double* w_std;     /* [N][A] Welford state of the readings; all NULL: no standardisation */
    float* x;                     /* [N][A][11] policy input of the step (rs_collect_pre); may be NULL for post_step / post_reset */
    float* xb;                    /* [N][A][11] policy input of the bootstrap round, or NULL */
    float* reward_used;           /* [N][A] or NULL */
    uint8_t* over; uint8_t* cut; uint8_t* boot;                        /* [N] */
    int64_t* pf_episode; int64_t* pf_calls; int64_t* episodes_begun;   /* [N] or NULL */
    int64_t* t;                   /* [1] device-side step counter, or NULL */
    /* copies of what rs_store_rows / rs_epoch_stats read of the env's output rows, taken by rs_collect_post_step so that rs_reset (which
     * rewrites those rows and moves the sources of the cut envs) may run beside the bootstrap round on another stream; all or none NULL */
    const uint8_t* env_oob;       /* [N][A] info.out_of_bounds */
    const int32_t* env_src_x; const int32_t* env_src_y;               /* [N] */
    uint8_t* done_copy; uint8_t* oob_copy;                            /* [N][A] */
    int32_t* src_copy;            /* [2][N] */
    int64_t* complete_len;        /* [N] or NULL: t + 1 of the env's last episode end so far in the epoch (what sample() of the 'cnn' update
                                   * counts, algos/multiagent/ppo.py:754-764); set by rs_collect_post_step where over != 0 */
} rs_collect_state;
int rs_collect_pre(const rs_collect_state* c, rs_stream_t stream);
int rs_collect_post_step(const rs_collect_state* c, int32_t epoch_ended, rs_stream_t stream);
int rs_collect_post_reset(const rs_collect_state* c, int32_t reset_hidden, rs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RADSEARCH_H */
