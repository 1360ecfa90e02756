/*
 * brl_hip.h — C-ABI of libbrl_hip.so: the MI355X-native (gfx950) bridge-bidding
 * environment + PPO-rollout hot path of harukaki/brl.
 *
 * The reference has no FFI layer: its boundary for this path is the Python API of
 * pgx.bridge_bidding plus src/{utils,roll_out,duplicate,gae}.py.  Each entry point below
 * names the reference interface it replaces (file:line under /root/reference).  The
 * Python host mirror in brl_amd/ binds these with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error (BRL_E_*); the message is in the
 *     thread-local brl_last_error().
 *   - unless a parameter says "host", every pointer is a DEVICE pointer on the handle's
 *     GPU, owned by the caller (e.g. torch tensors: tensor.data_ptr()).  The library owns
 *     only the handle, its device copy of the double-dummy LUT (plus the packed hand words it
 *     derives from the keys at upload, 32 B per row) and a small constant table.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and the compute entry points never
 *     synchronise (pass torch.cuda.current_stream().cuda_stream).  The three functions WITHOUT a stream
 *     argument that change what launches read — brl_create, brl_set_lut, brl_set_rng (when the values change) —
 *     call hipDeviceSynchronize first: kernels in flight may still read the old table / key.  Every entry point
 *     makes the handle's device current (hipSetDevice).
 *   - a handle is not thread-safe; use one per (device, stream).
 *   - per-table state is caller-owned and opaque: BRL_STATE_WORDS x uint64 per table
 *     (128 B, bit-packed: DESIGN.md "Data layout").  state_in == state_out is allowed.
 *   - batch arrays are batch-major [n, ...]; rollout outputs are time-major [T, n, ...]
 *     exactly like the reference's traj_batch (src/roll_out.py:105-107).
 *   - bool arrays are 1 byte per element (0/1), like the reference's jnp.bool_ arrays.
 */
#ifndef BRL_HIP_H
#define BRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BRL_STATE_WORDS 16  /* uint64 words per table */
#define BRL_OBS_SIZE 480    /* env.observation_shape == (480,)  ppo.py:241 */
#define BRL_NUM_ACTIONS 38  /* 0 Pass, 1 X, 2 XX, 3.. bids      src/duplicate.py:9-12 */

#define BRL_OK 0
#define BRL_E_ARG (-1)     /* bad argument */
#define BRL_E_HIP (-2)     /* a HIP runtime call failed */
#define BRL_E_NOLUT (-3)   /* a reset was requested but the handle has no LUT */

typedef struct brl_handle brl_handle;

const char *brl_last_error(void);
/* The version of the EXPORTED SET: the number of the build round in which a symbol was last added, removed or changed (6 now).
 * Under ONE version a symbol's signature and meaning never change.  Version 6 against version 5: added brl_mlp_gemm_x3 and
 * brl_mlp_gemm_x3_workspace, brl_mlp_gemm_x3_group (fp32 products on the bf16 matrix pipe at fp32-grade error: the large-batch forward
 * layers, the step's weight gradients as one launch), brl_split_planes, brl_linear_x3p (the large-batch inference layer on pre-split
 * operands); 54 symbols.
 * Version 5 against version 4 — the boundary is the path, the
 * experiment log (profiles/r04/r04_experiments.txt) keeps what was measured and dropped:
 *   removed (fusions of the PPO step that measured no faster): brl_mlp_gemm_bwd_pair, brl_mlp_gemm_fwd_heads,
 *     brl_ppo_heads_loss_parts, brl_adam_clip_fin_gather_defer, brl_mlp_gemm_adam, brl_adam_apply_range;
 *   removed (superseded forms): brl_ppo_loss_heads, brl_mb_gather, brl_relu_bwd_colsum, brl_adam_clip, brl_ppo_heads_loss,
 *     brl_adam_clip_gather (the multi-rank Adam: replaced by the two below);
 *   added: brl_adam_shard_norm, brl_adam_shard_apply (clip + Adam on a rank's slices of the bucketed flat buffers);
 *     brl_fair_chain, brl_mlp_gemm_group, brl_bias_finalize_rows (the FAIR network's step as five launches), brl_fair_forward.
 *     49 symbols. */
int brl_version(void);

/* BridgeBidding(dds_results_table_path)  — ppo.py:303, pgx.bridge_bidding.BridgeBidding.
 * lut_keys/lut_values: HOST pointers, int32 [lut_len,4] each, pgx packing (key: one word
 * per suit S,H,D,C, 13 base-4 digits = owner seat; value: one word per declarer seat, 5
 * hex digits = tricks in C,D,H,S,NT).  lut_len may be 0 (explicit deals only). */
int brl_create(int device, const int32_t *lut_keys, const int32_t *lut_values, int64_t lut_len,
               brl_handle **out);
/* LUT rotation — ppo.py:525-549 swaps the hash table; same arguments as brl_create.  Synchronises the device; a
 * table of the same length is copied over the old one (device addresses unchanged). */
int brl_set_lut(brl_handle *h, const int32_t *lut_keys, const int32_t *lut_values, int64_t lut_len);
int brl_destroy(brl_handle *h);

/* Seed of the counter-based RNG (Philox4x32-10) and the global index of this handle's
 * table 0 (rank * num_envs when env shards are spread over GPUs).  Replaces the PRNGKey
 * plumbing of ppo.py:314-333 / src/utils.py:49.  brl_init_random / brl_step / brl_rollout_random take the key by
 * value at launch; brl_policy_step[_at] read it (and the LUT) from a device-resident mirror, so a hipGraph replay of
 * a captured policy sub-step follows later brl_set_rng / brl_set_lut calls. */
int brl_set_rng(brl_handle *h, uint64_t seed, uint64_t env_offset);

/* jax.vmap(env.init)(keys)  — ppo.py:305,318.  Deals board number `board_ctr0` of every
 * table's stream: uniform LUT row, dealer, vulnerabilities, one of the 8 team-preserving
 * seatings. */
int brl_init_random(brl_handle *h, uint64_t *state, int64_t n, uint32_t board_ctr0, void *stream);

/* Explicit deals — the fields _duplicate_init copies (src/duplicate.py:120-128).
 * hand int32 [n,52] (13 pgx card ids per seat N,E,S,W), dealer int32 [n], vul_ns/vul_ew
 * uint8 [n], shuffled_players int32 [n,4] (seat -> player id), tricks uint8 [n,20]
 * ([declarer seat][C,D,H,S,NT]). */
int brl_init_from_deals(brl_handle *h, uint64_t *state, int64_t n, const int32_t *hand,
                        const int32_t *dealer, const uint8_t *vul_ns, const uint8_t *vul_ew,
                        const int32_t *shuffled_players, const uint8_t *tricks, void *stream);

/* env.step(state, action) — src/utils.py:44 (pgx core.Env.step); with autoreset != 0 it is
 * auto_reset(env.step, env.init) — src/utils.py:9-58.  action int32 [n].
 * Optional outputs for the NEW state (any may be NULL): obs uint8 [n,480] and
 * mask uint8 [n,38] of the new current player, rewards float [n,4] by player id,
 * terminated uint8 [n], current_player int32 [n]. */
int brl_step(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
             const int32_t *action, int autoreset, uint8_t *obs, uint8_t *mask, float *rewards,
             uint8_t *terminated, int32_t *current_player, void *stream);

/* _observe(state, player_id) — src/duplicate.py:6,134; player_id int32 [n] or NULL for
 * state.current_player.  obs uint8 [n,480]; mask uint8 [n,38] = state.legal_action_mask
 * (either may be NULL). */
int brl_observe(brl_handle *h, const uint64_t *state, int64_t n, const int32_t *player_id,
                uint8_t *obs, uint8_t *mask, void *stream);

/* State attribute access (pgx State fields).  Every pointer may be NULL. */
typedef struct brl_fields {
  int32_t *current_player;        /* [n]    */
  uint8_t *terminated;            /* [n]    */
  float *rewards;                 /* [n,4]  */
  int32_t *step_count;            /* [n]    */
  int32_t *turn;                  /* [n]    */
  int32_t *dealer;                /* [n]    */
  uint8_t *vul_ns;                /* [n]    */
  uint8_t *vul_ew;                /* [n]    */
  int32_t *shuffled_players;      /* [n,4]  */
  int32_t *last_bid;              /* [n]    */
  int32_t *last_bidder;           /* [n] player id, -1 none */
  uint8_t *call_x;                /* [n]    */
  uint8_t *call_xx;               /* [n]    */
  int32_t *pass_num;              /* [n]    */
  int32_t *first_denomination_ns; /* [n,5] seat, -1 none */
  int32_t *first_denomination_ew; /* [n,5]  */
  int32_t *hand;                  /* [n,52] ascending card ids per seat */
  uint8_t *tricks;                /* [n,20] */
  int32_t *lut_idx;               /* [n] -1 for explicit deals */
  uint32_t *board_ctr;            /* [n]    */
  uint8_t *illegal;               /* [n] an illegal action was taken on this table */
} brl_fields;
int brl_get_fields(brl_handle *h, const uint64_t *state, int64_t n, const brl_fields *out, void *stream);

/* Transition — src/roll_out.py:13-20; time-major [T,n,...].  Any pointer may be NULL. */
typedef struct brl_transition {
  uint8_t *done;               /* [T,n]      */
  int32_t *action;             /* [T,n]      */
  float *value;                /* [T,n]      */
  float *reward;               /* [T,n]  rewards[actor] / reward_scale */
  float *log_prob;             /* [T,n]      */
  uint8_t *obs;                /* [T,n,480]  */
  uint8_t *legal_action_mask;  /* [T,n,38]   */
} brl_transition;

/* roll_out with the uniform-random masked policy, T-loop fused into one launch —
 * src/roll_out.py:49-108 with auto_reset(env.step, env.init) (src/utils.py:9-58) and
 * normal_step (substeps=1, src/utils.py:249-254) or the 4-sub-step competitive macro-step
 * with all seats random (substeps=4, src/utils.py:69-128).  state is updated in place.
 * draw_base: index of the first action draw (advance by T*substeps between calls).
 * last_obs uint8 [n,480] / last_mask uint8 [n,38]: observation and legal mask of the post-rollout
 * state, i.e. runner_state's last_obs of src/roll_out.py:95-102 (either may be NULL).
 * terminated_count: device int64 accumulated like src/roll_out.py:85 (may be NULL).
 * Alignment: state and every output array 16-byte aligned (done: 4-byte) — the kernels store 16 bytes per lane;
 * BRL_E_ARG otherwise (also for brl_rollout_random_gae's advantages / targets). */
int brl_rollout_random(brl_handle *h, uint64_t *state, int64_t n, int num_steps, int substeps,
                       uint32_t draw_base, float reward_scale, const brl_transition *out,
                       uint8_t *last_obs, uint8_t *last_mask, int64_t *terminated_count, void *stream);

/* brl_rollout_random (substeps 1) + brl_gae of the SAME trajectory in ONE launch: with the random policy the value column
 * is 0, so the scan of src/gae.py:20-39 needs nothing the launch does not produce itself besides last_val float [n]
 * (the critic's value of the post-rollout observation: zeros for a policy without a critic).  advantages / targets float
 * [num_steps,n], bit-identical to brl_gae on the Transition this call writes.  One launch for 1 <= num_steps <= 40,
 * n % 32 == 0 and every Transition column; any other shape (done / value / reward columns required) runs as
 * brl_rollout_random followed by brl_gae on the same stream. */
int brl_rollout_random_gae(brl_handle *h, uint64_t *state, int64_t n, int num_steps, uint32_t draw_base,
                           float reward_scale, const brl_transition *out, uint8_t *last_obs, uint8_t *last_mask,
                           int64_t *terminated_count, const float *last_val, float gamma, float gamma_lambda,
                           float *advantages, float *targets, void *stream);

/* One policy sub-step: masked categorical over `logits` float [n,38] for the current
 * player (mode bit 0 clear: sample, src/roll_out.py:79-81 / src/utils.py:83-85; set: arg-max,
 * src/utils.py:157,174 / src/evaluation.py:135; mode bit 1 set: the UNMASKED categorical of the
 * illegal-action-penalty policy, src/roll_out.py:33-39 — an illegal draw ends the board with the pgx
 * penalty), then auto_reset(env.step) when autoreset != 0.  Uses draw index `draw` of each table's action stream.
 * Outputs (any may be NULL): action int32 [n], log_prob float [n] (log-softmax over the legal —
 * or, unmasked, all — actions at the chosen action), then as brl_step.  rewards_acc float [n,4] and
 * terminated_acc uint8 [n], when given, are ACCUMULATED (+=, |=) — src/utils.py:126-127. */
int brl_policy_step(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
                    const float *logits, int mode, uint32_t draw, int autoreset, int32_t *action,
                    float *log_prob, uint8_t *obs, uint8_t *mask, float *rewards_acc,
                    uint8_t *terminated_acc, int32_t *current_player, void *stream);

/* The same, generalised for loops that run without the host and for logits that are a slice of a wider matrix:
 *  - logits_stride: elements between the rows of consecutive tables (38 = dense; 39 when the actor and critic heads
 *    are one GEMM and the logits are its first 38 columns);
 *  - draw_base (device pointer, may be NULL): the draw index is *draw_base + draw_offset, read by the kernel when it
 *    runs — for scans captured once in a hipGraph and replayed (the arguments of a captured launch are frozen, the
 *    counter is advanced by another node of the graph); the reference's jitted scan has no host in its loop either
 *    (src/roll_out.py:63-108). */
int brl_policy_step_at(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
                       const float *logits, int64_t logits_stride, int mode, const uint32_t *draw_base,
                       uint32_t draw_offset, int autoreset, int32_t *action, float *log_prob, uint8_t *obs,
                       uint8_t *mask, float *rewards_acc, uint8_t *terminated_acc, int32_t *current_player,
                       void *stream);

/* Per-macro-step bookkeeping folded into a sub-step launch (every member optional: zero / NULL = off) — what the scan body
 * of src/roll_out.py:63-103 does around the four sub-steps of src/utils.py:69-128, without a launch of its own. */
typedef struct brl_macro_ext {
  int32_t first;             /* != 0: first sub-step of a macro-step: rewards_acc / terminated_acc are OVERWRITTEN */
  int32_t last;              /* != 0: last sub-step: the members below marked (last) are written after accumulation */
  const void *value_in;      /* value_out[i] = value_in[i * value_stride]: the critic output -> Transition.value[t] (:76) */
  int64_t value_stride;
  float *value_out;
  uint8_t *done_out;         /* (last) [n]  = terminated_acc                                   (src/roll_out.py:85, G2) */
  float *reward_out;         /* (last) [n]  = rewards_acc[i, actor[i]] / reward_scale          (src/roll_out.py:90, G1) */
  const int32_t *actor;      /*        [n]  player id that acted in sub-step 1 (src/roll_out.py:72) */
  float reward_scale;
  int32_t obs_fmt;           /* obs_cast element type: 0 float, 1 bf16, 2 fp16 */
  int64_t *terminated_count; /* (last) [1] += sum_i terminated_acc[i]                           (src/roll_out.py:85) */
  void *obs_cast;            /* [n,480]: the new observation as the next forward's input (`astype`, src/roll_out.py:75) */
  int32_t in_fmt;            /* element type of `logits` AND value_in as the GEMM wrote them: 0 float, 1 bf16, 2 fp16 (with 1 / 2
                                the `logits` argument points at 2-byte elements; strides stay in elements) */
  int32_t reserved;
  /* head_h != NULL: the launch forms the 38 logits + the value ITSELF — the policy heads `actor(x), critic(x)` (src/models.py:30-33)
   * on x = head_h [n, head_ldh] (bf16 / fp16, the last hidden layer's output), head_w [39, head_hidden] (same type: actor rows,
   * then the critic row), head_b float [39], fp32 accumulation — instead of reading `logits` / value_in (ignored; `logits` may be
   * NULL).  head_hidden % 32 == 0; needs 4 tables per wave (the default). */
  const void *head_h;
  int64_t head_ldh;
  const void *head_w;
  const float *head_b;
  int32_t head_hidden;
  int32_t head_fmt;          /* 1 bf16, 2 fp16 */
  /* head_part != NULL (instead of head_h): the heads arrive as PARTIAL products of brl_linear_act_heads — logits / value [i][j] =
   * head_b[j] + sum over p < head_nparts, in order, of head_part[p * head_part_stride + i * head_part_ld + j] (float);
   * head_part_ld >= 39.  Needs 4 tables per wave (the default). */
  const float *head_part;
  int64_t head_part_stride;
  int32_t head_part_ld;
  int32_t head_nparts;
} brl_macro_ext;

/* brl_policy_step_at + brl_macro_ext (ext may be NULL). */
int brl_policy_step_ex(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
                       const float *logits, int64_t logits_stride, int mode, const uint32_t *draw_base,
                       uint32_t draw_offset, int autoreset, int32_t *action, float *log_prob, uint8_t *obs,
                       uint8_t *mask, float *rewards_acc, uint8_t *terminated_acc, int32_t *current_player,
                       const brl_macro_ext *ext, void *stream);

/* Observation bytes -> network input: `last_obs.astype(jnp.float32)` (src/roll_out.py:75) and its low-precision
 * variants.  obs uint8 [n,480] (0/1); out [n,480] of float (fmt 0), bf16 (fmt 1) or fp16 (fmt 2). */
int brl_obs_cast(brl_handle *h, const uint8_t *obs, int64_t n, void *out, int fmt, void *stream);
/* The same for the rows `rows[0..m)` of obs only: out [m,480], out row r = obs row rows[r] (an evaluator forwards the boards
 * that are still playing). */
int brl_obs_cast_rows(brl_handle *h, const uint8_t *obs, const int64_t *rows, int64_t m, void *out, int fmt, void *stream);
/* The evaluators' loop condition `~state.terminated.all()` (src/evaluation.py:120-122, 153-178) as data, without a host round
 * trip: *finished = number of boards with terminated != 0; live[0 .. n - *finished) = the indices of the others, ascending
 * (entries behind them are left untouched).  Either output may be NULL.  `finished` may be pinned host memory (hipHostMalloc /
 * a pinned torch tensor): the count is then stored where the host reads it — no copy launch in between.  tag >= 0 (< 2^31):
 * *finished = (tag << 32) | count as ONE 64-bit store released at system scope: the host can poll the word while the stream
 * runs on (a fresh tag says which launch wrote it) instead of waiting for an event behind the launch — an event between two
 * iterations of an evaluator costs the stream ~6 us.  tag < 0: the plain count. */
int brl_live_index(brl_handle *h, const uint8_t *terminated, int64_t n, int64_t *live, int64_t *finished, int64_t tag,
                   void *stream);

/* One hidden layer of the policy network in 16-bit inference precision — `hk.Linear(1024)` + `jax.nn.relu`,
 * src/models.py:23-33, as called per env.step by src/roll_out.py:73-84 / src/evaluation.py:52-60:
 *     y[m, n_out] = act(x[m, k] w[n_out, k]^T + bias),  x / w / y bf16 (fmt 1) or fp16 (fmt 2), fp32 accumulation, bias fp32.
 * OPT-IN precision (narrower than the reference's fp32; `inference_dtype`), fp32 stays with the library GEMM.
 * w is nn.Linear's own [n_out, k] layout; row strides ldx / ldw / ldy in elements (% 8 == 0); n_out % 128 == 0; k % 8 == 0
 * (k = 480 is fine: nothing beyond column k of x or w is used); relu != 0 applies max(., 0); 16-byte aligned pointers. */
int brl_linear_act(brl_handle *h, const void *x, int64_t ldx, const void *w, int64_t ldw, const float *bias, void *y,
                 int64_t ldy, int64_t m, int n_out, int k, int relu, int fmt, void *stream);
/* The LAST hidden layer together with its share of the policy heads `actor(x), critic(x)` (src/models.py:30-33): as
 * brl_linear_act, and every 128-column tile p of y is multiplied, while the launch still holds it (rounded to `fmt` like the y
 * a separate head product would read), with columns 128 p .. 128 p + 127 of head_w [n_heads, ld_head_w] (same 16-bit type; n_heads
 * <= 48: 38 actor rows + the critic row): head_part[p * head_part_stride + i * head_part_ld + j] (float) = that partial product,
 * i < m, j < n_heads (head_part_ld % 4 == 0 and >= n_heads rounded up to a multiple of 4: the padding columns of a row may be
 * written, with zeros).  brl_policy_step_ex (brl_macro_ext.head_part) adds the n_out / 128 parts and the bias.  y may be NULL: the
 * layer's output is then not written at all. */
int brl_linear_act_heads(brl_handle *h, const void *x, int64_t ldx, const void *w, int64_t ldw, const float *bias, void *y,
                         int64_t ldy, int64_t m, int n_out, int k, int relu, int fmt, const void *head_w, int64_t ld_head_w,
                         int n_heads, float *head_part, int64_t head_part_ld, int64_t head_part_stride, void *stream);

/* calc_gae's reverse scan — src/gae.py:20-39.  done uint8 [T,n], value/reward float
 * [T,n], last_val float [n]; gamma_lambda = float32(gamma * gae_lambda). */
int brl_gae(brl_handle *h, const uint8_t *done, const float *value, const float *reward,
            const float *last_val, float gamma, float gamma_lambda, int T, int64_t n,
            float *advantages, float *targets, void *stream);

/* _imp_reward — src/duplicate.py:15-70.  a, b, out float [n,4]. */
int brl_imp_reward(brl_handle *h, const float *a, const float *b, float *out, int64_t n, void *stream);

/* Table_info — src/duplicate.py:138-144 (struct of arrays). */
typedef struct brl_table_info {
  uint8_t *terminated;  /* [n]   */
  float *rewards;       /* [n,4] */
  int32_t *last_bid;    /* [n]   */
  int32_t *last_bidder; /* [n]   */
  uint8_t *call_x;      /* [n]   */
  uint8_t *call_xx;     /* [n]   */
} brl_table_info;

/* duplicate_step(env.step) — src/duplicate.py:147-192; table_a/table_b updated in place.
 * Optional outputs as brl_step (rewards = the IMP vector on the step table B ends, else 0). */
int brl_duplicate_step(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
                       const int32_t *action, const brl_table_info *table_a,
                       const brl_table_info *table_b, uint8_t *obs, uint8_t *mask, float *rewards,
                       uint8_t *terminated, int32_t *current_player, void *stream);

/* Per-board counters of the evaluators' step log — src/evaluation.py:649-735 (duplicate) / :299-378 (single table);
 * index [board, team] with team 0 = players {0,1} (the "actor" side), team 1 = players {2,3}.  Any pointer may be NULL. */
typedef struct brl_eval_stats {
  float *illegal_prob_sum;  /* [n,2]    sum over the team's calls of the UNMASKED softmax mass on illegal actions */
  int32_t *step_count;      /* [n,2]    calls made by the team */
  int32_t *pass_count;      /* [n,2]    of which passes */
  int32_t *bid_count;       /* [n,2,35] how often the team made each bid (or 0/1 "made it at all": bid_set) */
} brl_eval_stats;

/* One iteration of an evaluator's loop with the networks' logits as input — src/evaluation.py:146-169 (simple
 * duplicate), :749-790 (duplicate with statistics), :380-403 (single table): per board the network of the team to act
 * (players {0,1}: logits_team1, else logits_team2; float [n,38] with row strides) plays masked_pi.mode(); boards that
 * are not finished update `stats`; then duplicate_step (table_a/table_b given, src/duplicate.py:147-192) or env.step
 * (both NULL); cum_return float [n] += rewards[0], rewards_sum float [n,4] += rewards (either may be NULL);
 * action_out int32 [n] (may be NULL).  Remaining outputs as brl_step. */
int brl_eval_step(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n,
                  const float *logits_team1, int64_t stride1, const float *logits_team2, int64_t stride2,
                  const brl_table_info *table_a, const brl_table_info *table_b, const brl_eval_stats *stats,
                  int bid_set, float *cum_return, float *rewards_sum, int32_t *action_out, uint8_t *obs,
                  uint8_t *mask, float *rewards, uint8_t *terminated, int32_t *current_player, void *stream);

/* brl_eval_step with ONE network per launch: only the boards whose player to act belongs to `acting_team` (0: players
 * {0,1}, 1: players {2,3}) take their greedy call from `logits`; the other unfinished boards wait (state, statistics and
 * accumulators untouched, action_out = -1), finished boards take their no-op step as always.  Alternating the team from
 * launch to launch plays every board exactly as brl_eval_step does (the calls are deterministic arg-maxes) with ONE forward
 * per iteration instead of the two the reference evaluates and selects from (src/evaluation.py:146-151): a board's teams
 * alternate call by call, so it waits at most one launch at its start and one at the table switch.
 * obs_f32 (optional, [n,480], 16-byte aligned): the new observation ALSO as float — the next forward's input
 * (`observation.astype(jnp.float32)`, src/evaluation.py:52), so that no cast launch sits in front of a full-batch forward. */
int brl_eval_step_team(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n, const float *logits,
                       int64_t stride, int acting_team, const brl_table_info *table_a, const brl_table_info *table_b,
                       const brl_eval_stats *stats, int bid_set, float *cum_return, float *rewards_sum, int32_t *action_out,
                       uint8_t *obs, uint8_t *mask, float *rewards, uint8_t *terminated, int32_t *current_player,
                       float *obs_f32, void *stream);

/* End-of-run histograms behind make_evaluate's log_info — src/evaluation.py:841-1031 (make_terminated_log,
 * make_contract_log) — as exact integer counts.  out: device int64 [BRL_EVAL_COUNTS], zeroed by the call:
 *   out[t*80 + 0] pass-outs at table t (0 = A, 1 = B; table_b may be NULL)
 *   out[t*80 + 1 + 2*team], out[t*80 + 2 + 2*team]  doubled / redoubled contracts declared by the team
 *   out[t*80 + 5 + team] / out[t*80 + 7 + team]      boards with rewards[0] >= 0 / < 0 by declaring team (the
 *                                                    reference's make_contract / down_contract, :951-984)
 *   out[t*80 + 9]                                    sum of rewards[0] (table scores are integers)
 *   out[t*80 + 10 + 35*team + bid]                   final contracts by declaring team and bid
 *   out[160 + 35*team + bid]                         sum over boards of stats.bid_count (bid_count may be NULL)
 *   out[230]                                         sum of _step_count of `state` (may be NULL) */
#define BRL_EVAL_COUNTS 231
int brl_eval_reduce(brl_handle *h, int64_t n, const brl_table_info *table_a, const brl_table_info *table_b,
                    const int32_t *bid_count, const uint64_t *state, int64_t *out, void *stream);

/* `_loss_fn` of the PPO update — src/update.py:90-167 — and its gradient w.r.t. the network's outputs in ONE launch
 * (the reference lets jax.value_and_grad differentiate ~60 elementwise ops; here torch differentiates only the GEMMs).
 * Inputs, one row per minibatch sample: logits float [batch,38] (row stride given), value float [batch], mask uint8
 * [batch,38] (traj_batch.legal_action_mask), action int32, old_value / old_log_prob / gae / targets float [batch].
 * masked != 0: the masked policy (src/update.py:12-16), else the unmasked one (:18-21); value_clipping: :48-60.
 * Outputs: dlogits float [batch,38] and dvalue float [batch] = d(loss_actor + vf_coef * value_loss - ent_coef * entropy)
 * / d(logits, value) with the 1/batch of the means folded in; partials float [ceil(batch/4), 8]: per-block sums of
 * (value-loss term, actor-loss term, entropy, approx-KL term, clipped?) — column sums / batch are the logged
 * statistics; illegal_probs float [batch,38] (may be NULL): softmax(logits) * ~mask (:136-137).
 * No handle: `device` is the HIP device the arrays live on. */
int brl_ppo_loss(int device, const float *logits, int64_t logits_stride, const float *value, const uint8_t *mask,
                 const int32_t *action, const float *old_value, const float *old_log_prob, const float *gae,
                 const float *targets, int64_t batch, float clip_eps, float vf_coef, float ent_coef, int masked,
                 int value_clipping, float *dlogits, float *dvalue, float *partials, float *illegal_probs, void *stream);

/* The logged statistics of that minibatch step (src/update.py:142-167), one launch: partials as written by brl_ppo_loss
 * for `batch` samples; gram = P^T P (row-major float [38,38], may be NULL) of P = illegal_probs, from which
 * `jnp.linalg.norm(P, ord=2) / 2` (:138-141) is obtained without an SVD (8 squarings + Rayleigh quotient).
 * out: float [8] = total_loss, value_loss, loss_actor, entropy, approx_kl, clipfrac, illegal-action norm / 2, 0. */
int brl_ppo_stats(int device, const float *partials, int64_t batch, const float *gram, float vf_coef, float ent_coef,
                  float *out, void *stream);

/* ---- one PPO minibatch step without the small launches (src/update.py:74-242; brl_amd/update.py::FusedMinibatch strings
 * these together with the GEMMs of the 4 x 1024 MLP and captures the step in one hipGraph) ------------------------------------- */

/* ---- the 39-column head (38 logits + value, src/models.py:30-33) of one PPO minibatch step as three entry points instead of
 * four library GEMMs with N or K = 39 and five small kernels (brl_amd/csrc/ppo_heads.hpp) -------------------------------------- */

/* heads = h head_w^T + head_b (h float [batch, ldh >= hidden], the last hidden layer's output; head_w float [39, hidden] = actor
 * rows then the critic row; head_b float [39]; hidden % 16 == 0), then `_loss_fn` (src/update.py:90-167) exactly as brl_ppo_loss
 * on that matrix (logits = columns 0..37, value = column 38).  TWO launches: the heads product split over K across workgroups
 * (ksplit ranges, 1..8; 4 at hidden = 1024) into head_parts float [ksplit, batch, 39] (scratch), then the loss on head_b + the
 * parts added in order.  Outputs: dheads float [batch,39] = d(total)/d(heads); partials float [ceil(batch / 4) * 8] for
 * brl_ppo_stats_gram; gram_partials (may be NULL) float [ceil(batch / 4) * 1444]: per 4-sample group, P^T P of its
 * illegal-action probabilities (src/update.py:136-141); heads_out (may be NULL) float [batch,39].  reward_scaling != 0: the
 * advantages are normalised over the minibatch first, (gae - mean) / (std + 1e-8) with jnp's ddof = 0 (src/update.py:31-44,118). */
int brl_ppo_heads_loss_split(int device, const float *h, int64_t ldh, const float *head_w, const float *head_b, int64_t hidden,
                       const uint8_t *mask, const int32_t *action, const float *old_value, const float *old_log_prob,
                       const float *gae, const float *targets, int64_t batch, float clip_eps, float vf_coef, float ent_coef,
                       int masked, int value_clipping, int reward_scaling, float *heads_out, float *dheads, float *partials,
                       float *gram_partials, float *head_parts, int ksplit, void *stream);

/* Backward of the head in one launch (what autograd does for `actor(x), critic(x)` and the activation under them):
 * dw_partials float [nsplit, 39 * hidden] and db_partials float [nsplit, 39]: d(heads)^T h and the column sums of d(heads)
 * per batch split of <= 64 rows (brl_bias_finalize_ex adds the splits in order: deterministic); dh float [batch, hidden] =
 * (d(heads) head_w) * act'(h) — act 0: ReLU (h > 0), 1: tanh (1 - h^2), src/models.py:16 —, i.e. the gradient w.r.t. the
 * last hidden layer's pre-activation, and tile_sums float [ceil(batch / 16), hidden]: its column sums per 16-row tile
 * (that layer's bias gradient, same layout as brl_act_bwd_colsum's scratch).  hidden % 256 == 0.
 * gram_sums != NULL: the same launches also add up brl_ppo_heads_loss_split's loss_partials float [ngroups,8] and gram_partials float
 * [ngroups,1444] (in group order) into row *row_index (device memory) of stat_sums float [rows,8] / gram_sums float [rows,1444]:
 * the statistics of a whole update are then formed by ONE brl_ppo_stats_rows at its end instead of a launch per step. */
/* (dw_partials = db_partials = NULL: only the activation-gradient role is launched — see brl_act_bwd_colsum_heads_dw) */
int brl_ppo_heads_bwd(int device, const float *dheads, const float *h, int64_t ldh, const float *head_w, int64_t batch,
                      int64_t hidden, int act, int nsplit, float *dw_partials, float *db_partials, float *dh, float *tile_sums,
                      const float *loss_partials, const float *gram_partials, int64_t ngroups, const int32_t *row_index,
                      float *stat_sums, float *gram_sums, void *stream);

/* The log row of one step from brl_ppo_heads_loss_split's outputs (written at row *row_index when given): partials float [npartials, 8], gram_partials float [ngram, 1444] (summed
 * in order).  The illegal-action norm comes from 4 squarings of G / trace by the whole block and 16 power-iteration steps with
 * that matrix (G^256 v, as brl_ppo_stats).  total_loss includes illegal_coef * norm / 2.  row_index may be NULL (row 0).  vec_out (may be NULL): float [40] = the top right
 * singular vector v1 [38], sigma_1, 0 — what the gradient of the norm needs (d sigma_1 / dP = u1 v1^T). */
int brl_ppo_stats_gram(int device, const float *partials, int64_t npartials, int64_t batch, const float *gram_partials,
                       int64_t ngram, float vf_coef, float ent_coef, float illegal_coef, float *out_rows, const int32_t *row_index,
                       float *vec_out, void *stream);

/* illegal_action_l2norm_coef != 0 (src/update.py:146-152): adds d(illegal_coef * sigma_1(P) / 2) / d(logits) to dheads float
 * [batch,39] in place — heads float [batch,39] as brl_ppo_heads_loss_split wrote them (heads_out), vec = brl_ppo_stats_gram's vec_out of
 * the same minibatch (the top right singular vector and sigma_1: d sigma_1 / dP = u1 v1^T with u1 = P v1 / sigma_1). */
int brl_ppo_illegal_grad(int device, const float *heads, const uint8_t *mask, const float *vec, float illegal_coef, int64_t batch,
                         float *dheads, void *stream);

/* The log rows of `rows` minibatch steps at once (one block per row): stat_sums / gram_sums as accumulated by
 * brl_ppo_heads_bwd; out_rows float [rows,8] as brl_ppo_stats_gram writes one. */
int brl_ppo_stats_rows(int device, const float *stat_sums, const float *gram_sums, int64_t rows, int64_t batch, float vf_coef,
                       float ent_coef, float illegal_coef, float *out_rows, void *stream);

/* `shuffled = take(batch, permutation)` + the slice of minibatch *mb_index (src/update.py:193-206): row perm[*mb_index * mbs + b]
 * of the flattened [T*N] trajectory `flat` (and of adv / targets) -> the static minibatch buffers; x0 float [mbs,480] =
 * obs.astype(float32) (src/update.py:95).  The launch reads its arguments from DEVICE memory: brl_mb_gather_bind writes them into
 * args_dev (256 bytes, stream-ordered: a one-thread launch, no host copy) once per update; brl_mb_gather_dev(args_dev) is then a
 * launch whose parameters never change, so it lives inside the captured minibatch step and follows a new trajectory /
 * permutation.  mb_index is DEVICE memory (the Adam entry points advance it): a captured graph walks through an epoch by itself.
 * nsteps = the minibatches `perm` holds: the Adam launch's gather of a minibatch >= nsteps is skipped. */
int brl_mb_gather_bind(int device, const brl_transition *flat, const float *adv, const float *targets, const int64_t *perm,
                       const int32_t *mb_index, int64_t mbs, float *x0, uint8_t *mask, int32_t *action, float *old_value,
                       float *old_log_prob, float *gae_out, float *targets_out, int64_t nsteps, void *args_dev, void *stream);
int brl_mb_gather_dev(int device, const void *args_dev, int64_t mbs, void *stream);

/* Backward of `h = act(z)` plus the bias gradient's partials of that layer (where brl_mlp_gemm's GATE_COLSUM epilogue does not do
 * it): dh [rows,ld] *= act'(h) in place (act 0: ReLU, 1: tanh) and the column sums of every 16-row tile into scratch float
 * [ceil(rows / 16), cols]; cols and ld multiples of 4. */
int brl_act_bwd_colsum(int device, float *dh, const float *h, int64_t rows, int64_t cols, int64_t ld, int act, float *scratch,
                       void *stream);
/* brl_act_bwd_colsum of ONE hidden layer with the weight-gradient role of brl_ppo_heads_bwd (dW_h / db_h partials + the step's
 * statistics / Gram sums; arguments as there) as extra workgroups of the same launch.  That role is not on the backward chain —
 * nothing needs its outputs before the sums at the end of the step — so it rides with a launch that is: call brl_ppo_heads_bwd
 * with dw_partials = db_partials = NULL (activation-gradient role only), then this for the first layer below the top. */
int brl_act_bwd_colsum_heads_dw(int device, float *dz, const float *hh, int64_t rows, int64_t cols, int64_t ld, int act,
                                float *scratch, const float *dheads, const float *h, int64_t ldh, int64_t batch, int64_t hidden,
                                int nsplit, float *dw_partials, float *db_partials, const float *loss_partials,
                                const float *gram_partials, int64_t ngroups, const int32_t *row_index, float *stat_sums,
                                float *gram_sums, void *stream);

/* The sums of several layers' tile partials in one launch: out[i][c] = sum_{t < tiles[i]} partials[i][t * cols[i] + c],
 * in order, for nseg <= 16 segments (bias gradients of the layers, and the head's weight / bias gradients from
 * brl_ppo_heads_bwd's batch splits). */
int brl_bias_finalize_ex(int device, int nseg, const float *const *partials, const int64_t *cols, const int64_t *tiles,
                         float *const *out, void *stream);
/* The same, with the segments i >= first_row_seg written as rows of a log: out[i] + *row_index * cols[i] (row_index: device
 * memory, e.g. the minibatch counter the Adam launches advance) — the step's statistics / Gram partials summed into row
 * *row_index of stat_sums [rows,8] / gram_sums [rows,1444] (what brl_ppo_stats_rows reads) by the launch that finishes the bias
 * gradients. */
int brl_bias_finalize_rows(int device, int nseg, const float *const *partials, const int64_t *cols, const int64_t *tiles,
                           float *const *out, int first_row_seg, const int32_t *row_index, void *stream);

/* ---- the "FAIR" network's minibatch step without its small launches (src/models.py:34-69 `actor_model_type == "FAIR"`: eleven
 * 200-wide hk.Linear's in four residual blocks, the observation concatenated back in front of the seventh, the two heads on the
 * last block's output; src/update.py:74-167 for the loss) — brl_amd/csrc/fair_chain.hpp, brl_amd/fused_update.py::FusedFair.
 * ONE launch: forward, `_loss_fn` and the whole backward chain of 16 samples per workgroup (rows are independent up to the weight
 * gradients; the activations stay in LDS, the weights stream from L2, products are fp32 MFMA tiles).  It leaves what the
 * weight-gradient products (library GEMMs) and brl_bias_finalize_rows need:
 *   inp   float [9][batch][200]  the inputs of the square layers 1,2,3,4,5,7,8,9,10 (in that order)
 *   dzs   float [9][batch][200]  d(loss)/d(their pre-activations)          -> dW_l = dzs[i]^T inp[i] (one batched product)
 *   cat6  float [batch][680] = [z5 | x0], dz6 float [batch][200]           -> dW_6 = dz6^T cat6
 *   dz0   float [batch][200]                                               -> dW_0 = dz0^T x0
 *   x4    float [batch][200], dheads float [batch][40] (column 39 = 0)     -> d(head_w) = dheads[:, :39]^T x4
 *   gates float [4][batch][200]  scratch (h2, h4, h8, h10: activation outputs the backward re-reads)
 *   tiles float [11][batch/16][200] + [batch/16][39]  per workgroup: column sums of dz_l (segments l = 0..10) and, behind them,
 *         of dheads = the bias gradients' partials (brl_bias_finalize_rows: tiles = batch / 16)
 *   partials float [batch/16][8], gram_partials float [batch/16][1444] (may be NULL): the statistics as brl_ppo_heads_loss_split's.
 * net: weights in nn.Linear's [out,in] layout (w[0] [200,480], w[6] [200,680], the others [200,200]); head_w [39,200] = actor rows,
 * then the critic row; head_b [39].  act: 0 ReLU, 1 tanh.  Loss arguments as brl_ppo_heads_loss_split.  batch % 16 == 0; every
 * array 16-byte aligned. */
typedef struct brl_fair_net {
  const float *w[11];
  const float *b[11];
  const float *head_w;
  const float *head_b;
} brl_fair_net;
typedef struct brl_fair_work {
  float *inp, *dzs, *gates, *cat6, *x4, *dz0, *dz6, *dheads, *tiles, *partials, *gram_partials;
} brl_fair_work;
/* `actor(x), critic(x)` of the FAIR network alone (src/models.py:34-69 as called per env.step by src/roll_out.py:73-76 and the
 * evaluators, src/evaluation.py:52-60): x float [rows,480] -> logits float [rows,38], value float [rows], ONE launch of the same
 * kernel's forward half (16 rows per workgroup; any number of rows).  head_w = actor rows then the critic row, contiguous. */
int brl_fair_forward(int device, const brl_fair_net *net, const float *x, int64_t rows, int act, float *logits, float *value,
                     void *stream);
int brl_fair_chain(int device, const brl_fair_net *net, const float *x0, const uint8_t *mask, const int32_t *action,
                   const float *old_value, const float *old_log_prob, const float *gae, const float *targets, int64_t batch,
                   float clip_eps, float vf_coef, float ent_coef, int masked, int value_clipping, int reward_scaling, int act,
                   const brl_fair_work *work, void *stream);

/* optax.chain(clip_by_global_norm(max_norm), adam(lr, eps)) (ppo.py:195-211) on FLAT fp32 buffers of n elements (n a multiple of
 * 4: pad with zeros), single rank, two launches.  Launch 1: *step (device float) += 1, *mb_index += 1 (may be NULL), the `nseg`
 * sums of partials (arguments as brl_bias_finalize_ex; they must be exactly the END of the gradient buffer g — the head's weight
 * gradient + every bias gradient in FusedMinibatch's flat layout, up to the buffer's zero padding) and the partial square sums of
 * the whole gradient.  Launch 2: the gradient scaled by min(1, max_norm / (|g| + 1e-6)) (max_norm <= 0: no clipping); m, v, p
 * updated with torch.optim.Adam's arithmetic; extra workgroups gather the NEXT minibatch (gather_args as written by
 * brl_mb_gather_bind, mbs its minibatch size; the first minibatch of an update is gathered by one brl_mb_gather_dev after the
 * bind; gather_args may be NULL).  lr_dev (may be NULL): the learning rate in device memory, used instead of `lr` (a captured
 * launch then follows ppo.py:186-192's linear schedule).  scratch: at least 1024 + sum over the segments of ceil(cols / 64)
 * floats.  norm_out (may be NULL): |g| before clipping. */
int brl_adam_clip_fin_gather(int device, float *p, float *g, float *m, float *v, int64_t n, float *step, float lr,
                             const float *lr_dev, float beta1, float beta2, float eps, float max_norm, float *scratch,
                             int64_t scratch_len, int32_t *mb_index, float *norm_out, const void *gather_args, int64_t mbs,
                             int nseg, const float *const *partials, const int64_t *cols, const int64_t *tiles, float *const *out,
                             void *stream);

/* The same under a process group (ppo.py has no counterpart: the reference is single-device; SURVEY section 8e: the gradient of
 * the flat buffers crosses xGMI once per minibatch).  The flat buffers are cut into `nbuckets` BUCKETS — one per collective —
 * and every bucket into `world` equal SLICES: bucket b = floats [off[b], off[b] + world * len[b]), slice r of it = what rank r
 * holds after a reduce-scatter of the bucket (off, len multiples of 4).
 *   brl_adam_shard_norm: partials[(r * nbuckets + b) * nsub + j] = the squares of sub-block j of slice (r, b) of grad_scale * g,
 *     for r in [rank_lo, rank_hi); *step += 1, *mb_index += 1 (may be NULL).  A rank that owns only its reduced slices computes
 *     its own row (rank_lo = rank, rank_hi = rank + 1) and an all-gather of world * nbuckets * nsub floats completes the array;
 *     a rank that holds the whole all-reduced gradient computes all rows — the SAME array either way.
 *   brl_adam_shard_apply: clip (norm = sqrt of the sum of ALL partials, added in a fixed order) + Adam on the slices of ranks
 *     [rank_lo, rank_hi); g is addressed like p (the reduce-scatter is in place); grad_scale = 1 / world after a SUM reduction;
 *     gather_args / mbs / lr_dev / norm_out as brl_adam_clip_fin_gather.
 * Sharded form of a step: reduce-scatter per bucket, norm of the own slices, all-gather of the partials, apply on the own slices,
 * all-gather of the parameters per bucket.  Replicated form: all-reduce, norm and apply with rank_lo = 0, rank_hi = world.
 * Both give bit-identical parameters (given the same reduced gradient). */
typedef struct brl_shard_geom {
  int32_t nbuckets;   /* 1..12 */
  int32_t world;
  int32_t nsub;       /* norm sub-blocks per slice */
  int32_t reserved;
  int64_t off[12];    /* first float of bucket b */
  int64_t len[12];    /* floats per SLICE of bucket b */
} brl_shard_geom;
int brl_adam_shard_norm(int device, const float *g, const brl_shard_geom *geom, int rank_lo, int rank_hi, float grad_scale,
                        float *partials, float *step, int32_t *mb_index, void *stream);
int brl_adam_shard_apply(int device, float *p, const float *g, float *m, float *v, const brl_shard_geom *geom, int rank_lo, int rank_hi,
                         const float *partials, const float *step, float lr, const float *lr_dev, float beta1, float beta2, float eps,
                         float max_norm, float grad_scale, float *norm_out, const void *gather_args, int64_t mbs, void *stream);

/* ---- round 4: the step's fp32 GEMMs with the elementwise work of src/update.py:74-242 in their epilogues -------------------
 * One fp32 product C[m,n] (row-major, ldc) on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, a k-ordered fma chain per
 * output), 64 x 64 or 64 x 32 output tiles (csrc/mlp_gemm.hpp; colsum / sqsum partials are per 64 ROWS either way; sqsum holds
 * one word per tile: size it for 64 x 32 tiles, ceil(m / 64) * ceil(n / 32)).  Layouts (what is contiguous in memory):
 *   BRL_GEMM_NT  c = a b^T         a [m,k] (lda), b [n,k] (ldb)          the forward pass: x W^T        (src/models.py:23-33)
 *   BRL_GEMM_NN  c = a b           a [m,k] (lda), b [k,n] (ldb)          dh = dz W                       (autograd of the same)
 *   BRL_GEMM_TN  c = a^T b         a [k,m] (lda), b [k,n] (ldb)          dW = dz^T h
 * Epilogues:
 *   BRL_GEMM_EPI_NONE
 *   BRL_GEMM_EPI_BIAS_ACT     (NT)  c = act(c + bias[n]); act 0 = ReLU, 1 = tanh                         (src/models.py:16)
 *   BRL_GEMM_EPI_GATE_COLSUM  (NN)  c = c * act'(gate[m,n]) with gate = the layer's forward OUTPUT (ReLU: gate > 0; tanh:
 *                                   1 - gate^2), and colsum [ceil(m / 64), n] = the column sums of every 64-row tile of what was
 *                                   stored (the bias gradient's partials: finish with brl_bias_finalize_ex, tiles = ceil(m / 64));
 *                                   colsum may be NULL
 *   BRL_GEMM_EPI_SQSUM        (TN)  sqsum [ceil(m / 64) * ceil(n / 64)] = the sum of squares of every output tile
 *                                   (clip_by_global_norm's partial sums, ppo.py:195-211)
 * n, lda, ldb, ldc (ldg) multiples of 4; k a multiple of 4 where it is an operand's contiguous index; where m is (TN), lda >= m
 * rounded up to 4 (whole 16-byte pieces are read; rows of c beyond m are not written); operands
 * below 2 GB.  Deterministic (no atomics).  Replaces torch.mm / torch.addmm + brl_act_bwd_colsum in FusedMinibatch. */
#define BRL_GEMM_NT 0
#define BRL_GEMM_NN 1
#define BRL_GEMM_TN 2
#define BRL_GEMM_EPI_NONE 0
#define BRL_GEMM_EPI_BIAS_ACT 1
#define BRL_GEMM_EPI_GATE_COLSUM 2
#define BRL_GEMM_EPI_SQSUM 3
int brl_mlp_gemm(int device, int layout, int epilogue, const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                 int64_t ldc, int64_t m, int64_t n, int64_t k, int act, const float *bias, const float *gate, int64_t ldg,
                 float *colsum, float *sqsum, void *stream);

/* The same product "bf16x3" (csrc/mlp_gemm_x3.hpp; config["inference_gemm"], opt-in): every fp32 operand as three exact bf16
 * pieces, six bf16 MFMA products per K step, fp32 accumulators by magnitude class — max |err| against float64 0.07-0.44 x
 * brl_mlp_gemm's on the same inputs, and NOT bit-identical to it; 128 x 128 tiles: for outputs of >= 256 of them (the policy
 * rollout's and the evaluators' forward layers, src/roll_out.py:49-108, src/models.py:23-33) 1.4 x brl_mlp_gemm's rate.  Arguments as
 * brl_mlp_gemm's; epilogues NONE, BIAS_ACT (NT), GATE_COLSUM (NN); k a multiple of 32 (whole chunks: every phase of the K loop is
 * branch-free; other k: brl_mlp_gemm); every pointer 16-byte aligned.  Outputs of fewer tiles may
 * divide K among several workgroups per tile: `workspace` (may be NULL: then never) = *bytes of brl_mlp_gemm_x3_workspace (0: none) of
 * device memory, 256-byte aligned, ZERO before the first call and owned by one stream at a time (partial tiles + a ticket per tile;
 * the last workgroup of a tile adds the partial tiles in K order: deterministic). */
int brl_mlp_gemm_x3_workspace(int64_t m, int64_t n, int64_t k, int64_t *bytes);
int brl_mlp_gemm_x3(int device, int layout, int epilogue, const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                    int64_t ldc, int64_t m, int64_t n, int64_t k, int act, const float *bias, const float *gate, int64_t ldg,
                    float *colsum, void *workspace, int64_t workspace_bytes, void *stream);

/* `count` <= 8 plain bf16x3 products of one layout as ONE launch, one K slice each (arguments as brl_mlp_gemm_group's; every pointer
 * 16-byte aligned, every k a multiple of 32): the DeepMind MLP's weight gradients dW_l = dz_l^T h_{l-1} of one minibatch step (src/update.py:74-242) — three
 * 1024 x 1024 outputs + one 1024 x 480 = 224 tiles, one per CU, instead of a batched library product + one more launch
 * (config["dw_gemm"] = "bf16x3"). */
int brl_mlp_gemm_x3_group(int device, int layout, int count, const float *const *a, const int64_t *lda, const float *const *b,
                          const int64_t *ldb, float *const *c, const int64_t *ldc, const int64_t *m, const int64_t *n,
                          const int64_t *k, void *stream);

/* The inference layer of LARGE batches on operands ALREADY split into bf16 planes (csrc/mlp_linear_x3p.hpp): in a rollout the weights
 * are constant over 128 forwards and an activation is split once by the launch that produces it, so the product itself is DMA -> LDS ->
 * MFMA with no vector work (brl_mlp_gemm_x3 splits every operand element once per tile that reads it).
 * brl_split_planes: x[n] (fp32, n a multiple of 4) -> planes[3][..]: plane p (0 hi, 1 mid, 2 lo; x == hi + mid + lo exactly) at
 * planes + p * plane_stride (uint16 elements). */
int brl_split_planes(int device, const float *x, int64_t n, uint16_t *planes, int64_t plane_stride, void *stream);
/* y[m, n] = (relu ? max(., 0) : .)(x[m, k] w[n, k]^T + bias[n]) — one `hk.Linear` (+ `jax.nn.relu`) of `forward_fn` (src/models.py:23-33) in
 * fp32-grade arithmetic: x as npx planes [npx][m][ldx] (plane stride sx; npx = 3, or 1 where x is exact in bf16: the 0/1 observation as the
 * step kernel writes it, brl_macro_ext.obs_cast with obs_fmt 1), w as three planes [3][n][ldw] (nn.Linear's layout; stride sw); outputs:
 * y fp32 [m][ldy] and / or y_planes [3][m][ldyp] (stride syp) — either may be NULL, not both.  n % 128 == 0, k % 32 == 0, leading
 * dimensions and strides of planes multiples of 8, every pointer 16-byte aligned. */
int brl_linear_x3p(int device, const uint16_t *x_planes, int npx, int64_t ldx, int64_t sx, const uint16_t *w_planes, int64_t ldw, int64_t sw,
                   const float *bias, int relu, float *y, int64_t ldy, uint16_t *y_planes, int64_t ldyp, int64_t syp, int64_t m, int64_t n,
                   int64_t k, void *stream);

/* `count` <= 16 plain products of one layout as ONE launch (arguments as brl_mlp_gemm's, one array element per product): the FAIR
 * network's eleven weight gradients dW_l = dz_l^T x_l (BRL_GEMM_TN; src/models.py:34-69's 200-wide layers are 0.08-0.5 GFLOP
 * each — a launch apiece costs more than the arithmetic). */
int brl_mlp_gemm_group(int device, int layout, int count, const float *const *a, const int64_t *lda, const float *const *b,
                       const int64_t *ldb, float *const *c, const int64_t *ldc, const int64_t *m, const int64_t *n,
                       const int64_t *k, void *stream);

/* brl_mlp_gemm(BRL_GEMM_NN, BRL_GEMM_EPI_GATE_COLSUM, dz, w, ...) of the hidden layer below the top with the weight-gradient
 * role of brl_ppo_heads_bwd (arguments as brl_act_bwd_colsum_heads_dw) as extra workgroups of the same launch: call
 * brl_ppo_heads_bwd with dw_partials = db_partials = NULL, then this as the first product of the dz chain. */
int brl_mlp_gemm_dh_heads_dw(int device, const float *dz, int64_t lddz, const float *w, int64_t ldw, float *out, int64_t ldo,
                             int64_t m, int64_t n, int64_t k, int act, const float *gate, int64_t ldg, float *colsum,
                             const float *dheads, const float *h, int64_t ldh, int64_t batch, int64_t hidden, int nsplit,
                             float *dw_partials, float *db_partials, const float *loss_partials, const float *gram_partials,
                             int64_t ngroups, const int32_t *row_index, float *stat_sums, float *gram_sums, void *stream);

/* ---- the policy network's fp32 forward for the boards an evaluator still plays, own kernels end to end (round 4) ------------
 * `actor(x), critic(x)` of src/models.py:23-33 ("DeepMind": nlayers x [hk.Linear(hidden) + activation], then the two heads) as
 * called per env.step by src/evaluation.py:52-60,146-151, for m selected rows: x[r] = float(obs[rows[r]]) (obs uint8 [.,480],
 * 0/1; rows NULL: row r itself), nlayers launches of brl_mlp_gemm's kernel (BRL_GEMM_NT + BRL_GEMM_EPI_BIAS_ACT; the weights
 * in nn.Linear's own [out, in] layout — nothing is transposed, copied or cached), one launch for the 38 + 1 heads that writes
 * row r's logits and value to out[rows[r] * ldo + 0..38] (the scatter back included).  One host call instead of the ~8 launches
 * through torch of the library path: an evaluator's small-batch iterations are bound by host launches.
 * scratch: m * (480 + 2 * hidden) floats, 16-byte aligned.  in_features must be 480, hidden % 4 == 0 and <= 1024. */
typedef struct brl_mlp_ref {
  int32_t nlayers;          /* hidden layers, 1..8 */
  int32_t act;              /* 0 = ReLU, 1 = tanh (src/models.py:16) */
  int64_t in_features;      /* 480 */
  int64_t hidden;
  const float *w[8];        /* w[l]: [hidden, in_l] row-major, in_0 = in_features, in_l = hidden */
  const float *b[8];        /* b[l]: [hidden] */
  const float *actor_w;     /* [38, hidden] */
  const float *actor_b;     /* [38] */
  const float *critic_w;    /* [1, hidden] */
  const float *critic_b;    /* [1] */
} brl_mlp_ref;
int brl_mlp_forward_rows(int device, const brl_mlp_ref *net, const uint8_t *obs, const int64_t *rows, int64_t m, float *scratch,
                         int64_t scratch_len, float *out, int64_t ldo, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BRL_HIP_H */
