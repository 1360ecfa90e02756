/*
 * oracle/bridge_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, deliberately un-clever restatement of the bridge-bidding hot path of
 * harukaki/brl (pgx.bridge_bidding.{init,step,observe} + src/utils.py auto_reset /
 * macro-step + src/roll_out.py + src/duplicate.py + src/gae.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this
 * file's shared object.  The product (brl_amd/) never imports, links or calls it.
 *
 * PARITY STATUS: **parity unpinned** against pgx==1.4.0 itself.  pgx is a third-party
 * dependency of the reference (requirements.txt:40) that is neither vendored under
 * /root/reference nor installable in the build container, so the auction state machine,
 * legal mask, scoring and LUT packing below restate pgx 1.4.0's published algorithm as
 * evidenced by the reference's own call sites (each function cites them).  What IS pinned
 * by reference-held data: the IMP table + 6 doctest KATs (src/duplicate.py:20-50), the
 * 480-bit observation layout (wb5/utils.py:15-52), the 14-call auction of
 * wb5/utils.py:61-64, the seat swap of src/duplicate.py:113-128, the score extremes
 * 3500 / 7600 (src/duplicate.py:36, ppo.py:174) and the 1000 double-dummy deals of
 * wb5/dataset_for_vs_wb5.json.  See tests/test_oracle_kat.py.
 *
 * State is kept UNPACKED, field for field like pgx's State dataclass (field names proven
 * by src/utils.py:36-52, src/duplicate.py:113-127,170-173, src/evaluation.py:465-467),
 * and the observation is rebuilt from the full bidding history on every call — a
 * different formulation from the HIP kernels' incremental bit-packed one, on purpose.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_NUM_ACTIONS 38
#define ORC_OBS 480
#define ORC_HIST 319

#define ACT_PASS 0 /* src/duplicate.py:9  */
#define ACT_X 1    /* src/duplicate.py:10 */
#define ACT_XX 2   /* src/duplicate.py:11 */
#define ACT_BID0 3 /* src/duplicate.py:12 */

typedef struct orc_state {
  int32_t current_player;
  int32_t terminated;
  int32_t truncated;
  int32_t step_count;
  int32_t turn;
  int32_t dealer;
  int32_t vul_ns;
  int32_t vul_ew;
  int32_t last_bid;    /* -1 .. 34 = (level-1)*5 + strain, strain C,D,H,S,NT (src/evaluation.py:1084-1106) */
  int32_t last_bidder; /* PLAYER ID, -1 if none (src/evaluation.py:465,493) */
  int32_t call_x;
  int32_t call_xx;
  int32_t pass_num;
  int32_t illegal;   /* build-side: an illegal action was taken on this table */
  int32_t mask_all;  /* build-side: legal_action_mask was forced to all-True at a terminal */
  int32_t lut_idx;   /* build-side: LUT row this board came from, -1 for explicit deals */
  uint32_t board_ctr; /* build-side: how many boards this env slot has dealt (RNG counter) */
  int32_t shuffled_players[4];    /* seat -> player id (src/duplicate.py:113-115) */
  int32_t first_denomination_ns[5]; /* SEAT that first named strain for N/S, -1 none */
  int32_t first_denomination_ew[5];
  float rewards[4]; /* indexed by PLAYER ID (src/roll_out.py:90) */
  int32_t hand[52]; /* 13 card ids per seat N,E,S,W (workspace/test_bridge_with_openspiel.py:80) */
  uint8_t tricks[20]; /* double-dummy tricks [declarer seat][strain C,D,H,S,NT] */
  uint8_t legal_action_mask[ORC_NUM_ACTIONS];
  uint8_t observation[ORC_OBS];
  int16_t bidding_history[ORC_HIST + 1];
} orc_state;

int orc_sizeof_state(void) { return (int)sizeof(orc_state); }

/* ------------------------------------------------------------------------------------
 * Counter-based RNG (build-side; JAX threefry streams are not reproducible here, see
 * SURVEY §7 "RNG").  Philox4x32-10 (Salmon et al., SC'11).
 * ---------------------------------------------------------------------------------- */
void orc_philox4x32(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
  uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
  uint32_t k0 = key_in[0], k1 = key_in[1];
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

#define STREAM_RESET 0x42524C52u  /* 'BRLR' */
#define STREAM_ACTION 0x42524C41u /* 'BRLA' */

static uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

/* the draw used to pick an action for (env, draw index) */
uint32_t orc_action_draw(uint64_t seed, uint64_t env_id, uint32_t draw) {
  uint32_t ctr[4] = {(uint32_t)env_id, draw >> 2, STREAM_ACTION, (uint32_t)(env_id >> 32)};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t out[4];
  orc_philox4x32(ctr, key, out);
  return out[draw & 3];
}

/* ------------------------------------------------------------------------------------
 * Card conventions.
 * pgx card id = suit*13 + rank, suit S,H,D,C, rank A,2..K [RECALL, SURVEY App. B];
 * observation index = rank*4 + suit, suit C,D,H,S, rank 2..A  (wb5/utils.py:18-19,
 * workspace/test_bridge_with_openspiel.py:101-104).
 * ---------------------------------------------------------------------------------- */
int orc_card_to_obs_index(int card) {
  int suit = card / 13, rank = card % 13;
  int os_suit = 3 - suit;
  int os_rank = (rank + 12) % 13;
  return os_rank * 4 + os_suit;
}

/* LUT key: 4 x int32, one per suit, 13 base-4 digits (most significant first) = owner
 * seat of card suit*13+j  (wb5/vis_pgx.py:13-24 builds exactly this from a PBN).  */
void orc_key_to_hand(const int32_t key[4], int32_t hand[52]) {
  int owner[52];
  for (int s = 0; s < 4; s++) {
    uint32_t k = (uint32_t)key[s];
    for (int j = 12; j >= 0; j--) {
      owner[s * 13 + j] = (int)(k & 3u);
      k >>= 2;
    }
  }
  int n = 0;
  for (int seat = 0; seat < 4; seat++) /* stable: ascending card id within a seat */
    for (int c = 0; c < 52; c++)
      if (owner[c] == seat) hand[n++] = c;
}

void orc_hand_to_key(const int32_t hand[52], int32_t key[4]) {
  int owner[52];
  for (int i = 0; i < 52; i++) owner[hand[i]] = i / 13;
  for (int s = 0; s < 4; s++) {
    uint32_t k = 0;
    for (int j = 0; j < 13; j++) k = k * 4u + (uint32_t)owner[s * 13 + j];
    key[s] = (int32_t)k;
  }
}

/* LUT value: 4 x int32, one per declarer seat N,E,S,W; 5 hex digits, most significant
 * first, strain order C,D,H,S,NT [RECALL: pgx indexes dds_tricks[declarer*5 + last_bid%5];
 * docstring KAT 4160=0x01040 -> 0,1,0,4,0 ; 904605=0xDCD9D -> 13,12,13,9,13]. */
void orc_value_to_tricks(const int32_t value[4], uint8_t tricks[20]) {
  for (int seat = 0; seat < 4; seat++) {
    uint32_t v = (uint32_t)value[seat];
    for (int d = 4; d >= 0; d--) {
      tricks[seat * 5 + d] = (uint8_t)(v & 15u);
      v >>= 4;
    }
  }
}

void orc_tricks_to_value(const uint8_t tricks[20], int32_t value[4]) {
  for (int seat = 0; seat < 4; seat++) {
    uint32_t v = 0;
    for (int d = 0; d < 5; d++) v = v * 16u + tricks[seat * 5 + d];
    value[seat] = (int32_t)v;
  }
}

/* ------------------------------------------------------------------------------------
 * Seat helpers.
 * ---------------------------------------------------------------------------------- */
static int player_position(const orc_state *s, int player) {
  if (player < 0) return -1;
  for (int seat = 0; seat < 4; seat++)
    if (s->shuffled_players[seat] == player) return seat;
  return -1;
}

/* ------------------------------------------------------------------------------------
 * A3  _observe(state, player_id) — layout from wb5/utils.py:15-52.
 * ---------------------------------------------------------------------------------- */
void orc_observe(const orc_state *s, int player_id, uint8_t obs[ORC_OBS]) {
  memset(obs, 0, ORC_OBS);
  int pos = player_position(s, player_id);
  /* [0:4] vulnerability from the observer's side (wb5/utils.py:15-16) */
  int we = (pos == 0 || pos == 2) ? s->vul_ns : s->vul_ew;
  int they = (pos == 0 || pos == 2) ? s->vul_ew : s->vul_ns;
  obs[0] = (uint8_t)!we;
  obs[1] = (uint8_t)we;
  obs[2] = (uint8_t)!they;
  obs[3] = (uint8_t)they;
  /* [4:428] history (wb5/utils.py:28-46) */
  uint8_t *h = obs + 4;
  int last_bid = -1; /* wb5 uses 1-based ints with 0 = none; same thing shifted */
  for (int i = 0; i < ORC_HIST; i++) {
    int call = s->bidding_history[i];
    if (call < 0) break;
    int rel = ((i + s->dealer) % 4 + (4 - pos)) % 4;
    if (call >= ACT_BID0) {
      last_bid = call - ACT_BID0;
      h[4 + last_bid * 12 + rel] = 1;
    } else if (call == ACT_PASS) {
      if (last_bid < 0) h[rel] = 1;
    } else if (call == ACT_X) {
      if (last_bid >= 0) h[4 + last_bid * 12 + 4 + rel] = 1;
    } else if (call == ACT_XX) {
      if (last_bid >= 0) h[4 + last_bid * 12 + 8 + rel] = 1;
    }
  }
  /* [428:480] own cards (wb5/utils.py:18-26) */
  for (int i = 0; i < 13; i++) obs[428 + orc_card_to_obs_index(s->hand[pos * 13 + i])] = 1;
}

/* ------------------------------------------------------------------------------------
 * A1  init (explicit-deal form; fields = what _duplicate_init copies, src/duplicate.py:113-128)
 * ---------------------------------------------------------------------------------- */
void orc_init_explicit(orc_state *s, const int32_t hand[52], int dealer, int vul_ns, int vul_ew,
                       const int32_t shuffled[4], const uint8_t tricks[20]) {
  memset(s, 0, sizeof(*s));
  memcpy(s->hand, hand, sizeof(s->hand));
  memcpy(s->tricks, tricks, 20);
  s->dealer = dealer;
  s->vul_ns = vul_ns;
  s->vul_ew = vul_ew;
  for (int i = 0; i < 4; i++) s->shuffled_players[i] = shuffled[i];
  s->current_player = shuffled[dealer]; /* src/duplicate.py:115 */
  s->last_bid = -1;
  s->last_bidder = -1;
  s->lut_idx = -1;
  for (int i = 0; i < 5; i++) s->first_denomination_ns[i] = s->first_denomination_ew[i] = -1;
  for (int i = 0; i <= ORC_HIST; i++) s->bidding_history[i] = -1;
  for (int a = 0; a < ORC_NUM_ACTIONS; a++) s->legal_action_mask[a] = 1;
  s->legal_action_mask[ACT_X] = 0; /* src/duplicate.py:116-119 */
  s->legal_action_mask[ACT_XX] = 0;
  orc_observe(s, s->current_player, s->observation);
}

/* seeded form: uniform LUT row, dealer, vulnerabilities, one of the 8 team-preserving
 * seatings.  Draw layout is the build's own (DESIGN.md "RNG"). */
void orc_init_random(orc_state *s, uint64_t seed, uint64_t env_id, uint32_t board_ctr,
                     const int32_t *lut_keys, const int32_t *lut_values, int64_t lut_len) {
  uint32_t ctr[4] = {(uint32_t)env_id, board_ctr, STREAM_RESET, (uint32_t)(env_id >> 32)};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t r[4];
  orc_philox4x32(ctr, key, r);
  int64_t idx = (int64_t)mulhi32(r[0], (uint32_t)lut_len);
  int dealer = (int)(r[1] & 3u);
  int vul_ns = (int)((r[1] >> 2) & 1u);
  int vul_ew = (int)((r[1] >> 3) & 1u);
  int arr = (int)((r[1] >> 4) & 7u);
  int a0 = arr & 1, b0 = 2 + ((arr >> 1) & 1), ns_is_a = (arr >> 2) & 1;
  int a1 = 1 - a0, b1 = 5 - b0;
  int32_t sh[4];
  if (ns_is_a) { sh[0] = a0; sh[1] = b0; sh[2] = a1; sh[3] = b1; }
  else         { sh[0] = b0; sh[1] = a0; sh[2] = b1; sh[3] = a1; }
  int32_t hand[52];
  uint8_t tricks[20];
  orc_key_to_hand(lut_keys + idx * 4, hand);
  orc_value_to_tricks(lut_values + idx * 4, tricks);
  orc_init_explicit(s, hand, dealer, vul_ns, vul_ew, sh, tricks);
  s->lut_idx = (int32_t)idx;
  s->board_ctr = board_ctr;
}

/* ------------------------------------------------------------------------------------
 * A4  duplicate-bridge score for the declaring side (public laws).
 * denomination 0..4 = C,D,H,S,NT ; level 1..7 ; trick = tricks taken by declarer.
 * ---------------------------------------------------------------------------------- */
int orc_score(int denomination, int level, int vul, int call_x, int call_xx, int trick) {
  int need = level + 6;
  if (trick < need) {
    int u = need - trick;
    if (!call_x && !call_xx) return -(vul ? 100 : 50) * u;
    int pen;
    if (vul) {
      pen = 200 + (u > 1 ? 300 * (u - 1) : 0);
    } else {
      if (u == 1) pen = 100;
      else if (u == 2) pen = 300;
      else if (u == 3) pen = 500;
      else pen = 500 + 300 * (u - 3);
    }
    if (call_xx) pen *= 2;
    return -pen;
  }
  int per = (denomination <= 1) ? 20 : 30;
  int base = per * level + (denomination == 4 ? 10 : 0);
  int m = call_xx ? 4 : (call_x ? 2 : 1);
  int points = base * m;
  int score = points;
  score += (points >= 100) ? (vul ? 500 : 300) : 50;
  if (level == 6) score += vul ? 750 : 500;
  if (level == 7) score += vul ? 1500 : 1000;
  if (call_xx) score += 100;
  else if (call_x) score += 50;
  int over = trick - need;
  if (call_xx) score += over * (vul ? 400 : 200);
  else if (call_x) score += over * (vul ? 200 : 100);
  else score += over * per;
  return score;
}

/* terminal reward vector by PLAYER ID (workspace/test_bridge_with_openspiel.py:118-123) */
static void make_reward(orc_state *s) {
  for (int i = 0; i < 4; i++) s->rewards[i] = 0.0f;
  if (s->last_bid < 0) return; /* pass-out (src/evaluation.py:465-467) */
  int denomination = s->last_bid % 5;
  int level = s->last_bid / 5 + 1;
  int bidder_pos = player_position(s, s->last_bidder);
  int ns = (bidder_pos == 0 || bidder_pos == 2);
  int declarer = ns ? s->first_denomination_ns[denomination] : s->first_denomination_ew[denomination];
  int vul = ns ? s->vul_ns : s->vul_ew;
  int trick = s->tricks[declarer * 5 + denomination];
  int score = orc_score(denomination, level, vul, s->call_x, s->call_xx, trick);
  for (int seat = 0; seat < 4; seat++) {
    int seat_ns = (seat == 0 || seat == 2);
    s->rewards[s->shuffled_players[seat]] = (float)((seat_ns == ns) ? score : -score);
  }
}

/* ------------------------------------------------------------------------------------
 * A2  env.step (pgx core.Env.step wrapper + bridge _step)
 * ---------------------------------------------------------------------------------- */
static int is_partner_pos(int p, int q) { return ((p - q) & 1) == 0; }

void orc_step(orc_state *s, int action) {
  if (s->terminated || s->truncated) {
    /* terminated state stepped again: zero rewards, nothing else moves */
    for (int i = 0; i < 4; i++) s->rewards[i] = 0.0f;
  } else {
    int illegal = !s->legal_action_mask[action];
    int cur = s->current_player;
    int cur_pos = player_position(s, cur);
    s->step_count += 1;
    s->bidding_history[s->turn] = (int16_t)action;
    if (action == ACT_PASS) {
      s->pass_num += 1;
    } else if (action == ACT_X) {
      s->call_x = 1;
      s->pass_num = 0;
    } else if (action == ACT_XX) {
      s->call_xx = 1;
      s->pass_num = 0;
    } else {
      s->last_bid = action - ACT_BID0;
      s->last_bidder = cur;
      int d = s->last_bid % 5;
      if (cur_pos == 0 || cur_pos == 2) {
        if (s->first_denomination_ns[d] < 0) s->first_denomination_ns[d] = cur_pos;
      } else {
        if (s->first_denomination_ew[d] < 0) s->first_denomination_ew[d] = cur_pos;
      }
      s->call_x = 0;
      s->call_xx = 0;
      s->pass_num = 0;
      for (int a = ACT_BID0; a <= action; a++) s->legal_action_mask[a] = 0;
    }
    int term = (s->last_bid < 0 && s->pass_num == 4) || (s->last_bid >= 0 && s->pass_num == 3);
    if (term) {
      s->terminated = 1;
      make_reward(s);
    } else {
      int next_pos = (s->dealer + s->turn + 1) % 4;
      s->current_player = s->shuffled_players[next_pos];
      s->turn += 1;
      int has_bid = s->last_bidder >= 0;
      int bidder_pos = player_position(s, s->last_bidder);
      int own = has_bid && is_partner_pos(bidder_pos, next_pos);
      s->legal_action_mask[ACT_X] = (uint8_t)(has_bid && !own && !s->call_x && !s->call_xx);
      s->legal_action_mask[ACT_XX] = (uint8_t)(has_bid && own && s->call_x && !s->call_xx);
      for (int i = 0; i < 4; i++) s->rewards[i] = 0.0f;
    }
    if (illegal) { /* [RECALL] pgx: offender -1, the other three +1 each ... *(n-1) */
      for (int i = 0; i < 4; i++) s->rewards[i] = 3.0f;
      s->rewards[cur] = -1.0f;
      s->terminated = 1;
      s->illegal = 1;
    }
  }
  if (s->terminated) {
    for (int a = 0; a < ORC_NUM_ACTIONS; a++) s->legal_action_mask[a] = 1;
    s->mask_all = 1;
  }
  orc_observe(s, s->current_player, s->observation);
}

/* ------------------------------------------------------------------------------------
 * A5  auto_reset(step, init)  (src/utils.py:33-56)
 * ---------------------------------------------------------------------------------- */
void orc_auto_reset_step(orc_state *s, int action, uint64_t seed, uint64_t env_id,
                         const int32_t *lut_keys, const int32_t *lut_values, int64_t lut_len) {
  if (s->terminated || s->truncated) { /* src/utils.py:34-43 */
    s->step_count = 0;
    s->terminated = 0;
    s->truncated = 0;
    for (int i = 0; i < 4; i++) s->rewards[i] = 0.0f;
  }
  orc_step(s, action); /* src/utils.py:44 */
  if (s->terminated || s->truncated) { /* src/utils.py:45-55 */
    int term = s->terminated, trunc = s->truncated, ill = s->illegal;
    float r[4];
    memcpy(r, s->rewards, sizeof(r));
    uint32_t next = s->board_ctr + 1;
    orc_init_random(s, seed, env_id, next, lut_keys, lut_values, lut_len);
    s->terminated = term;
    s->truncated = trunc;
    s->illegal = ill;
    memcpy(s->rewards, r, sizeof(r));
  }
}

/* uniform-random legal action: k-th legal action in ascending order, k = mulhi(draw, n) */
int orc_random_action(const orc_state *s, uint32_t draw, int *n_legal_out) {
  int n = 0;
  for (int a = 0; a < ORC_NUM_ACTIONS; a++) n += s->legal_action_mask[a];
  int k = (int)mulhi32(draw, (uint32_t)n);
  if (n_legal_out) *n_legal_out = n;
  for (int a = 0; a < ORC_NUM_ACTIONS; a++) {
    if (s->legal_action_mask[a]) {
      if (k == 0) return a;
      k--;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------
 * A7  roll_out with a uniform-random masked policy (src/roll_out.py:63-107), BASELINE
 * config 1/2.  `substeps` = 1 is normal_step (src/utils.py:249-254); 4 is the competitive
 * macro-step (src/utils.py:69-128) with all four seats drawing uniformly.
 * Outputs are time-major [T,N,...] (G7).  value = 0 (no critic in the random policy).
 * ---------------------------------------------------------------------------------- */
void orc_rollout_random(orc_state *states, int64_t n_envs, int num_steps, int substeps,
                        uint64_t seed, uint64_t env_offset, uint32_t draw_base,
                        const int32_t *lut_keys, const int32_t *lut_values, int64_t lut_len,
                        float reward_scale,
                        uint8_t *obs, uint8_t *mask, int32_t *action, float *logp, float *value,
                        float *reward, uint8_t *done, int64_t *terminated_count) {
  int64_t tc = 0;
  /* envs are independent: e outer / t inner gives the same result as the reference's
   * t outer / vmap inner, and lets the cpu_baseline leg use every host core. */
#pragma omp parallel for reduction(+ : tc) schedule(static)
  for (int64_t e = 0; e < n_envs; e++) {
    for (int t = 0; t < num_steps; t++) {
      orc_state *s = &states[e];
      int64_t row = (int64_t)t * n_envs + e;
      uint64_t env_id = env_offset + (uint64_t)e;
      int actor = s->current_player; /* src/roll_out.py:72 */
      if (obs) memcpy(obs + row * ORC_OBS, s->observation, ORC_OBS);       /* G4 */
      if (mask) memcpy(mask + row * ORC_NUM_ACTIONS, s->legal_action_mask, ORC_NUM_ACTIONS);
      float rsum[4] = {0, 0, 0, 0};
      int term_any = 0;
      for (int k = 0; k < substeps; k++) {
        int n_legal;
        uint32_t draw = draw_base + (uint32_t)(t * substeps + k);
        int a = orc_random_action(s, orc_action_draw(seed, env_id, draw), &n_legal);
        if (k == 0) {
          if (action) action[row] = a;
          if (logp) logp[row] = (float)(-log((double)n_legal)); /* log-prob of a uniform pick */
          if (value) value[row] = 0.0f;
        }
        orc_auto_reset_step(s, a, seed, env_id, lut_keys, lut_values, lut_len);
        for (int i = 0; i < 4; i++) rsum[i] += s->rewards[i]; /* src/utils.py:126 */
        term_any |= s->terminated;                            /* src/utils.py:127 */
      }
      memcpy(s->rewards, rsum, sizeof(rsum));
      s->terminated = term_any;
      if (reward) reward[row] = rsum[actor] / reward_scale; /* G1, src/roll_out.py:90 */
      if (done) done[row] = (uint8_t)term_any;              /* G2 */
      tc += term_any;
    }
  }
  if (terminated_count) *terminated_count += tc;
}

/* ------------------------------------------------------------------------------------
 * A9  GAE (src/gae.py:20-39).  Time-major [T,N].
 * ---------------------------------------------------------------------------------- */
void orc_gae(const uint8_t *done, const float *value, const float *reward, const float *last_val,
             float gamma, float gamma_lambda, int T, int64_t N, float *adv, float *tgt) {
  for (int64_t n = 0; n < N; n++) {
    float gae = 0.0f, next_value = last_val[n];
    for (int t = T - 1; t >= 0; t--) {
      int64_t i = (int64_t)t * N + n;
      float nd = 1.0f - (float)done[i];
      float delta = reward[i] + gamma * next_value * nd - value[i]; /* src/gae.py:28 */
      gae = delta + gamma_lambda * nd * gae;                        /* src/gae.py:29 */
      adv[i] = gae;
      tgt[i] = gae + value[i]; /* src/gae.py:39 */
      next_value = value[i];
    }
  }
}

/* ------------------------------------------------------------------------------------
 * A10  _imp_reward (src/duplicate.py:15-70)
 * ---------------------------------------------------------------------------------- */
static const float IMP_LIST[24] = {20,  50,  90,   130,  170,  220,  270,  320,  370,  430,  500,  600,
                                   750, 900, 1100, 1300, 1500, 1750, 2000, 2250, 2500, 3000, 3500, 4000};

void orc_imp_reward(const float a[4], const float b[4], float out[4]) {
  float d = a[0] + b[0];
  float win = d >= 0 ? 1.0f : -1.0f; /* src/duplicate.py:52-54 */
  float ad = fabsf(d);
  int imp = 0;
  while (imp < 24 && ad >= IMP_LIST[imp]) imp++; /* src/duplicate.py:56-69 */
  float v = (float)imp * win;
  out[0] = v; out[1] = v; out[2] = -v; out[3] = -v; /* src/duplicate.py:70 */
}

/* ------------------------------------------------------------------------------------
 * A11  _duplicate_init / duplicate_init (src/duplicate.py:73-135)
 * ---------------------------------------------------------------------------------- */
void orc_duplicate_init(const orc_state *src, orc_state *dst) {
  static const int ix[4] = {1, 0, 3, 2}; /* src/duplicate.py:113 */
  int32_t sh[4], hand[52];
  uint8_t tricks[20];
  for (int i = 0; i < 4; i++) sh[i] = src->shuffled_players[ix[i]];
  memcpy(hand, src->hand, sizeof(hand));
  memcpy(tricks, src->tricks, 20);
  int dealer = src->dealer, vn = src->vul_ns, ve = src->vul_ew, li = src->lut_idx;
  uint32_t bc = src->board_ctr;
  orc_init_explicit(dst, hand, dealer, vn, ve, sh, tricks);
  dst->lut_idx = li;
  dst->board_ctr = bc;
}

/* A12  Table_info + duplicate_step (src/duplicate.py:138-192) */
typedef struct orc_table_info {
  int32_t terminated;
  float rewards[4];
  int32_t last_bid;
  int32_t last_bidder;
  int32_t call_x;
  int32_t call_xx;
} orc_table_info;

int orc_sizeof_table_info(void) { return (int)sizeof(orc_table_info); }

static void snapshot(const orc_state *s, orc_table_info *t) {
  t->terminated = s->terminated;
  memcpy(t->rewards, s->rewards, sizeof(t->rewards));
  t->last_bid = s->last_bid;
  t->last_bidder = s->last_bidder;
  t->call_x = s->call_x;
  t->call_xx = s->call_xx;
}

void orc_duplicate_step(orc_state *s, int action, orc_table_info *A, orc_table_info *B) {
  orc_step(s, action); /* src/duplicate.py:149 */
  orc_state stepped = *s;
  int a_done = A->terminated, b_done = B->terminated;
  if (!a_done && stepped.terminated) { /* src/duplicate.py:151-155 */
    orc_duplicate_init(&stepped, s);
  }
  if (a_done && stepped.terminated && !b_done) { /* src/duplicate.py:157-163 */
    *s = stepped;
    orc_imp_reward(A->rewards, stepped.rewards, s->rewards);
  } else {
    for (int i = 0; i < 4; i++) s->rewards[i] = 0.0f;
  }
  if (stepped.terminated && a_done && !b_done) snapshot(&stepped, B); /* :165-176 */
  if (!a_done && stepped.terminated) snapshot(&stepped, A);           /* :177-188 */
}

/* ------------------------------------------------------------------------------------
 * Batched conveniences for the Python test harness / cpu_baseline.
 * ---------------------------------------------------------------------------------- */
void orc_init_random_batch(orc_state *states, int64_t n, uint64_t seed, uint64_t env_offset,
                           const int32_t *lut_keys, const int32_t *lut_values, int64_t lut_len) {
  for (int64_t e = 0; e < n; e++)
    orc_init_random(&states[e], seed, env_offset + (uint64_t)e, 0, lut_keys, lut_values, lut_len);
}

void orc_step_batch(orc_state *states, int64_t n, const int32_t *action, int autoreset, uint64_t seed,
                    uint64_t env_offset, const int32_t *lut_keys, const int32_t *lut_values,
                    int64_t lut_len) {
  for (int64_t e = 0; e < n; e++) {
    if (autoreset)
      orc_auto_reset_step(&states[e], action[e], seed, env_offset + (uint64_t)e, lut_keys, lut_values,
                          lut_len);
    else
      orc_step(&states[e], action[e]);
  }
}

void orc_duplicate_step_batch(orc_state *states, int64_t n, const int32_t *action, orc_table_info *A,
                              orc_table_info *B) {
  for (int64_t e = 0; e < n; e++) orc_duplicate_step(&states[e], action[e], &A[e], &B[e]);
}
