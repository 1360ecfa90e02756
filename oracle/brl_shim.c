/*
 * brl_shim.c — the CPU oracle behind the SAME C symbols as libbrl_hip.so (include/brl_hip.h), so that a
 * parity scenario written against the C-ABI can be pointed at either library.  TEST INFRASTRUCTURE ONLY, like the rest
 * of oracle/: nothing under brl_amd/ may load it (tests/test_capi_cpu.py checks), and it takes HOST pointers.
 *
 * The ABI's per-table state is opaque to callers ("BRL_STATE_WORDS x uint64, caller-owned"): here word 0 of a table's 16
 * words holds 1 + the index of an `orc_state` in the handle's arena, the other words are unused.  A call with
 * state_out != state_in allocates fresh arena entries (the arena only grows; brl_destroy frees it).
 * Entry points the oracle has no counterpart for (policy sub-step over logits, dtype casts, evaluator / PPO kernels)
 * are exported and return BRL_E_ARG with a message.
 */
#include "bridge_oracle.c"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../include/brl_hip.h"

struct brl_handle {
  int32_t *keys, *values;
  int64_t lut_len;
  uint64_t seed, env_offset;
  orc_state *arena;
  int64_t used, cap;
};

static _Thread_local char g_err[256] = "";
static int fail(const char *msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return BRL_E_ARG;
}
const char *brl_last_error(void) { return g_err; }
int brl_version(void) { return 6; }   /* the exported set of include/brl_hip.h, version 6 */

static int set_lut(brl_handle *h, const int32_t *keys, const int32_t *values, int64_t len) {
  free(h->keys);
  free(h->values);
  h->keys = h->values = NULL;
  h->lut_len = 0;
  if (len > 0) {
    if (!keys || !values) return fail("lut_keys / lut_values are NULL with lut_len > 0");
    h->keys = (int32_t *)malloc((size_t)len * 16);
    h->values = (int32_t *)malloc((size_t)len * 16);
    memcpy(h->keys, keys, (size_t)len * 16);
    memcpy(h->values, values, (size_t)len * 16);
    h->lut_len = len;
  }
  return BRL_OK;
}

int brl_create(int device, const int32_t *lut_keys, const int32_t *lut_values, int64_t lut_len, brl_handle **out) {
  (void)device;
  if (!out || lut_len < 0) return fail("bad argument");
  brl_handle *h = (brl_handle *)calloc(1, sizeof(brl_handle));
  int rc = set_lut(h, lut_keys, lut_values, lut_len);
  if (rc) {
    free(h);
    return rc;
  }
  *out = h;
  return BRL_OK;
}
int brl_set_lut(brl_handle *h, const int32_t *k, const int32_t *v, int64_t len) { return h ? set_lut(h, k, v, len) : fail("handle"); }
int brl_destroy(brl_handle *h) {
  if (h) {
    free(h->keys);
    free(h->values);
    free(h->arena);
    free(h);
  }
  return BRL_OK;
}
int brl_set_rng(brl_handle *h, uint64_t seed, uint64_t env_offset) {
  if (!h) return fail("bad argument: handle");
  h->seed = seed;
  h->env_offset = env_offset;
  return BRL_OK;
}

static orc_state *fresh(brl_handle *h, uint64_t *slot_word) {
  if (h->used == h->cap) {
    h->cap = h->cap ? 2 * h->cap : 1024;
    h->arena = (orc_state *)realloc(h->arena, (size_t)h->cap * sizeof(orc_state));
  }
  *slot_word = (uint64_t)(++h->used);
  return &h->arena[h->used - 1];
}
static orc_state *at(brl_handle *h, const uint64_t *state, int64_t e) { return &h->arena[state[e * BRL_STATE_WORDS] - 1]; }

/* state_out entry for table e: the input entry when in place, else a fresh copy of it */
static orc_state *out_entry(brl_handle *h, const uint64_t *in, uint64_t *out, int64_t e) {
  if (in == out) return at(h, in, e);
  uint64_t w;
  int64_t src = (int64_t)in[e * BRL_STATE_WORDS] - 1;
  orc_state *d = fresh(h, &w); /* may move the arena: index the source afterwards */
  *d = h->arena[src];
  memset(out + e * BRL_STATE_WORDS, 0, BRL_STATE_WORDS * 8);
  out[e * BRL_STATE_WORDS] = w;
  return d;
}

static void outputs(const orc_state *s, int64_t e, uint8_t *obs, uint8_t *mask, float *rewards, uint8_t *terminated,
                    int32_t *current_player) {
  if (obs) memcpy(obs + e * BRL_OBS_SIZE, s->observation, BRL_OBS_SIZE);
  if (mask) memcpy(mask + e * BRL_NUM_ACTIONS, s->legal_action_mask, BRL_NUM_ACTIONS);
  if (rewards) memcpy(rewards + e * 4, s->rewards, 16);
  if (terminated) terminated[e] = (uint8_t)s->terminated;
  if (current_player) current_player[e] = s->current_player;
}

int brl_init_random(brl_handle *h, uint64_t *state, int64_t n, uint32_t board_ctr0, void *stream) {
  (void)stream;
  if (!h || !state || n < 0) return fail("bad argument");
  if (h->lut_len == 0) return BRL_E_NOLUT;
  for (int64_t e = 0; e < n; e++) {
    uint64_t w;
    orc_state *s = fresh(h, &w);
    orc_init_random(s, h->seed, h->env_offset + (uint64_t)e, board_ctr0, h->keys, h->values, h->lut_len);
    memset(state + e * BRL_STATE_WORDS, 0, BRL_STATE_WORDS * 8);
    state[e * BRL_STATE_WORDS] = w;
  }
  return BRL_OK;
}

int brl_init_from_deals(brl_handle *h, uint64_t *state, int64_t n, const int32_t *hand, const int32_t *dealer,
                        const uint8_t *vul_ns, const uint8_t *vul_ew, const int32_t *shuffled_players,
                        const uint8_t *tricks, void *stream) {
  (void)stream;
  if (!h || !state || !hand || !dealer || !vul_ns || !vul_ew || !shuffled_players || !tricks) return fail("NULL input array");
  for (int64_t e = 0; e < n; e++) {
    uint64_t w;
    orc_state *s = fresh(h, &w);
    orc_init_explicit(s, hand + e * 52, dealer[e], vul_ns[e] != 0, vul_ew[e] != 0, shuffled_players + e * 4, tricks + e * 20);
    memset(state + e * BRL_STATE_WORDS, 0, BRL_STATE_WORDS * 8);
    state[e * BRL_STATE_WORDS] = w;
  }
  return BRL_OK;
}

int brl_step(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n, const int32_t *action,
             int autoreset, uint8_t *obs, uint8_t *mask, float *rewards, uint8_t *terminated, int32_t *current_player,
             void *stream) {
  (void)stream;
  if (!h || !state_in || !state_out || !action) return fail("NULL state / action");
  if (autoreset && h->lut_len == 0) return BRL_E_NOLUT;
  for (int64_t e = 0; e < n; e++) {
    orc_state *s = out_entry(h, state_in, state_out, e);
    if (autoreset)
      orc_auto_reset_step(s, action[e], h->seed, h->env_offset + (uint64_t)e, h->keys, h->values, h->lut_len);
    else
      orc_step(s, action[e]);
    outputs(s, e, obs, mask, rewards, terminated, current_player);
  }
  return BRL_OK;
}

int brl_observe(brl_handle *h, const uint64_t *state, int64_t n, const int32_t *player_id, uint8_t *obs, uint8_t *mask,
                void *stream) {
  (void)stream;
  if (!h || !state) return fail("state");
  for (int64_t e = 0; e < n; e++) {
    const orc_state *s = at(h, state, e);
    if (obs) orc_observe(s, player_id ? player_id[e] : s->current_player, obs + e * BRL_OBS_SIZE);
    if (mask) memcpy(mask + e * BRL_NUM_ACTIONS, s->legal_action_mask, BRL_NUM_ACTIONS);
  }
  return BRL_OK;
}

int brl_get_fields(brl_handle *h, const uint64_t *state, int64_t n, const brl_fields *F, void *stream) {
  (void)stream;
  if (!h || !state || !F) return fail("state / out");
  for (int64_t e = 0; e < n; e++) {
    const orc_state *s = at(h, state, e);
    if (F->current_player) F->current_player[e] = s->current_player;
    if (F->terminated) F->terminated[e] = (uint8_t)s->terminated;
    if (F->rewards) memcpy(F->rewards + e * 4, s->rewards, 16);
    if (F->step_count) F->step_count[e] = s->step_count;
    if (F->turn) F->turn[e] = s->turn;
    if (F->dealer) F->dealer[e] = s->dealer;
    if (F->vul_ns) F->vul_ns[e] = (uint8_t)s->vul_ns;
    if (F->vul_ew) F->vul_ew[e] = (uint8_t)s->vul_ew;
    if (F->shuffled_players) memcpy(F->shuffled_players + e * 4, s->shuffled_players, 16);
    if (F->last_bid) F->last_bid[e] = s->last_bid;
    if (F->last_bidder) F->last_bidder[e] = s->last_bidder;
    if (F->call_x) F->call_x[e] = (uint8_t)s->call_x;
    if (F->call_xx) F->call_xx[e] = (uint8_t)s->call_xx;
    if (F->pass_num) F->pass_num[e] = s->pass_num;
    if (F->first_denomination_ns) memcpy(F->first_denomination_ns + e * 5, s->first_denomination_ns, 20);
    if (F->first_denomination_ew) memcpy(F->first_denomination_ew + e * 5, s->first_denomination_ew, 20);
    if (F->hand) memcpy(F->hand + e * 52, s->hand, 208);
    if (F->tricks) memcpy(F->tricks + e * 20, s->tricks, 20);
    if (F->lut_idx) F->lut_idx[e] = s->lut_idx;
    if (F->board_ctr) F->board_ctr[e] = s->board_ctr;
    if (F->illegal) F->illegal[e] = (uint8_t)s->illegal;
  }
  return BRL_OK;
}

int brl_rollout_random(brl_handle *h, uint64_t *state, int64_t n, int num_steps, int substeps, uint32_t draw_base,
                       float reward_scale, const brl_transition *out, uint8_t *last_obs, uint8_t *last_mask,
                       int64_t *terminated_count, void *stream) {
  (void)stream;
  if (!h || !state || !out) return fail("state / out");
  if (h->lut_len == 0) return BRL_E_NOLUT;
  orc_state *tmp = (orc_state *)malloc((size_t)n * sizeof(orc_state));
  for (int64_t e = 0; e < n; e++) tmp[e] = *at(h, state, e);
  int64_t tc = 0;
  orc_rollout_random(tmp, n, num_steps, substeps, h->seed, h->env_offset, draw_base, h->keys, h->values, h->lut_len,
                     reward_scale, out->obs, out->legal_action_mask, out->action, out->log_prob, out->value, out->reward,
                     out->done, &tc);
  for (int64_t e = 0; e < n; e++) {
    *at(h, state, e) = tmp[e];
    outputs(&tmp[e], e, last_obs, last_mask, NULL, NULL, NULL);
  }
  free(tmp);
  if (terminated_count) *terminated_count += tc;
  return BRL_OK;
}

int brl_gae(brl_handle *h, const uint8_t *done, const float *value, const float *reward, const float *last_val,
            float gamma, float gamma_lambda, int T, int64_t n, float *advantages, float *targets, void *stream) {
  (void)h;
  (void)stream;
  orc_gae(done, value, reward, last_val, gamma, gamma_lambda, T, n, advantages, targets);
  return BRL_OK;
}

int brl_imp_reward(brl_handle *h, const float *a, const float *b, float *out, int64_t n, void *stream) {
  (void)h;
  (void)stream;
  for (int64_t e = 0; e < n; e++) orc_imp_reward(a + 4 * e, b + 4 * e, out + 4 * e);
  return BRL_OK;
}

static void load_info(const brl_table_info *T, int64_t e, orc_table_info *t) {
  t->terminated = T->terminated[e];
  memcpy(t->rewards, T->rewards + 4 * e, 16);
  t->last_bid = T->last_bid[e];
  t->last_bidder = T->last_bidder[e];
  t->call_x = T->call_x[e];
  t->call_xx = T->call_xx[e];
}
static void store_info(const orc_table_info *t, int64_t e, const brl_table_info *T) {
  T->terminated[e] = (uint8_t)t->terminated;
  memcpy(T->rewards + 4 * e, t->rewards, 16);
  T->last_bid[e] = t->last_bid;
  T->last_bidder[e] = t->last_bidder;
  T->call_x[e] = (uint8_t)t->call_x;
  T->call_xx[e] = (uint8_t)t->call_xx;
}

int brl_duplicate_step(brl_handle *h, const uint64_t *state_in, uint64_t *state_out, int64_t n, const int32_t *action,
                       const brl_table_info *table_a, const brl_table_info *table_b, uint8_t *obs, uint8_t *mask,
                       float *rewards, uint8_t *terminated, int32_t *current_player, void *stream) {
  (void)stream;
  if (!h || !state_in || !state_out || !action || !table_a || !table_b) return fail("NULL argument");
  for (int64_t e = 0; e < n; e++) {
    orc_state *s = out_entry(h, state_in, state_out, e);
    orc_table_info A, B;
    load_info(table_a, e, &A);
    load_info(table_b, e, &B);
    orc_duplicate_step(s, action[e], &A, &B);
    store_info(&A, e, table_a);
    store_info(&B, e, table_b);
    outputs(s, e, obs, mask, rewards, terminated, current_player);
  }
  return BRL_OK;
}

/* ---- no oracle counterpart: exported (with the header's signatures) so that one binding loads either library */
#define NOT_HERE(name) return fail(name ": not implemented by the oracle shim (no CPU counterpart)")
int brl_policy_step(brl_handle *h, const uint64_t *si, uint64_t *so, int64_t n, const float *lg, int mode, uint32_t draw,
                    int ar, int32_t *a, float *lp, uint8_t *o, uint8_t *m, float *r, uint8_t *t, int32_t *c, void *s) {
  (void)h; (void)si; (void)so; (void)n; (void)lg; (void)mode; (void)draw; (void)ar; (void)a; (void)lp; (void)o; (void)m; (void)r; (void)t; (void)c; (void)s;
  NOT_HERE("brl_policy_step");
}
int brl_policy_step_at(brl_handle *h, const uint64_t *si, uint64_t *so, int64_t n, const float *lg, int64_t ls, int mode,
                       const uint32_t *db, uint32_t d, int ar, int32_t *a, float *lp, uint8_t *o, uint8_t *m, float *r,
                       uint8_t *t, int32_t *c, void *s) {
  (void)h; (void)si; (void)so; (void)n; (void)lg; (void)ls; (void)mode; (void)db; (void)d; (void)ar; (void)a; (void)lp; (void)o; (void)m; (void)r; (void)t; (void)c; (void)s;
  NOT_HERE("brl_policy_step_at");
}
int brl_obs_cast(brl_handle *h, const uint8_t *obs, int64_t n, void *out, int fmt, void *s) {
  (void)h; (void)obs; (void)n; (void)out; (void)fmt; (void)s;
  NOT_HERE("brl_obs_cast");
}
int brl_linear_act(brl_handle *h, const void *x, int64_t ldx, const void *w, int64_t ldw, const float *bias, void *y,
                 int64_t ldy, int64_t m, int n_out, int k, int relu, int fmt, void *s) {
  (void)h; (void)x; (void)ldx; (void)w; (void)ldw; (void)bias; (void)y; (void)ldy; (void)m; (void)n_out; (void)k; (void)relu; (void)fmt; (void)s;
  NOT_HERE("brl_linear_act");
}
int brl_linear_act_heads(brl_handle *h, const void *x, int64_t ldx, const void *w, int64_t ldw, const float *bias, void *y,
                         int64_t ldy, int64_t m, int n_out, int k, int relu, int fmt, const void *head_w, int64_t ld_head_w,
                         int n_heads, float *head_part, int64_t head_part_ld, int64_t head_part_stride, void *s) {
  (void)h; (void)x; (void)ldx; (void)w; (void)ldw; (void)bias; (void)y; (void)ldy; (void)m; (void)n_out; (void)k; (void)relu; (void)fmt;
  (void)head_w; (void)ld_head_w; (void)n_heads; (void)head_part; (void)head_part_ld; (void)head_part_stride; (void)s;
  NOT_HERE("brl_linear_act_heads");
}
int brl_obs_cast_rows(brl_handle *h, const uint8_t *obs, const int64_t *rows, int64_t m, void *out, int fmt, void *s) {
  (void)h; (void)obs; (void)rows; (void)m; (void)out; (void)fmt; (void)s;
  NOT_HERE("brl_obs_cast_rows");
}
int brl_live_index(brl_handle *h, const uint8_t *terminated, int64_t n, int64_t *live, int64_t *finished, int64_t tag, void *s) {
  (void)h; (void)s;
  if (!terminated || (!live && !finished) || n <= 0) {
    snprintf(g_err, sizeof(g_err), "bad argument: brl_live_index (oracle shim)");
    return BRL_E_ARG;
  }
  int64_t k = 0;
  for (int64_t i = 0; i < n; i++)
    if (!terminated[i]) { if (live) live[k] = i; k++; }
  if (finished) *finished = tag >= 0 ? ((tag << 32) | (n - k)) : (n - k);
  return BRL_OK;
}
/* brl_mlp_gemm on the host: the plain definition (include/brl_hip.h), float64 accumulation — the checker of the MFMA kernel's
 * fp32 fma chains for small shapes (tests compare at 2e-4 * max|ref|); same argument rules as the library. */
int brl_mlp_gemm(int device, int layout, int epilogue, const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                 int64_t ldc, int64_t m, int64_t n, int64_t k, int act, const float *bias, const float *gate, int64_t ldg,
                 float *colsum, float *sqsum, void *s) {
  (void)device; (void)s;
  if (!a || !b || !c || m <= 0 || n <= 0 || k <= 0 || layout < 0 || layout > 2 || n % 4 || lda % 4 || ldb % 4 || ldc % 4 ||
      (layout != 2 && k % 4) || (act != 0 && act != 1) ||
      !(epilogue == 0 || (epilogue == 1 && layout == 0 && bias) || (epilogue == 2 && layout == 1 && gate) ||
        (epilogue == 3 && layout == 2 && sqsum))) {
    snprintf(g_err, sizeof(g_err), "bad argument: brl_mlp_gemm (oracle shim)");
    return BRL_E_ARG;
  }
  const int64_t tm = (m + 63) / 64, tn = (n + 63) / 64;
  if (epilogue == 3) for (int64_t t = 0; t < tm * tn; t++) sqsum[t] = 0.0f;
  if (epilogue == 2 && colsum) for (int64_t t = 0; t < tm * n; t++) colsum[t] = 0.0f;
  for (int64_t i = 0; i < m; i++)
    for (int64_t j = 0; j < n; j++) {
      double acc = 0.0;
      for (int64_t q = 0; q < k; q++) {
        const double av = layout == 2 ? a[q * lda + i] : a[i * lda + q];
        const double bv = layout == 0 ? b[j * ldb + q] : b[q * ldb + j];
        acc += av * bv;
      }
      if (epilogue == 1) { acc += bias[j]; acc = act == 0 ? (acc > 0 ? acc : 0) : tanh(acc); }
      if (epilogue == 2) { const double h = gate[i * ldg + j]; acc = act == 0 ? (h > 0 ? acc : 0) : acc * (1.0 - h * h); }
      const float v = (float)acc;
      c[i * ldc + j] = v;
      if (epilogue == 2 && colsum) colsum[(i / 64) * n + j] += v;
      if (epilogue == 3) sqsum[(i / 64) * tn + j / 64] += v * v;
    }
  return BRL_OK;
}
/* brl_mlp_gemm_x3 on the host: the SAME plain definition in float64 (the device kernel's three-piece operands and six products are an
 * implementation of the fp32 product, not a different function); no partial tiles here: the workspace is not touched */
int brl_mlp_gemm_x3_workspace(int64_t m, int64_t n, int64_t k, int64_t *bytes) { (void)m; (void)n; (void)k; if (bytes) *bytes = 0; return BRL_OK; }
int brl_mlp_gemm_x3(int device, int layout, int epilogue, const float *a, int64_t lda, const float *b, int64_t ldb, float *c,
                    int64_t ldc, int64_t m, int64_t n, int64_t k, int act, const float *bias, const float *gate, int64_t ldg,
                    float *colsum, void *workspace, int64_t workspace_bytes, void *s) {
  (void)workspace; (void)workspace_bytes;
  if (epilogue == BRL_GEMM_EPI_SQSUM || k % 32) {
    snprintf(g_err, sizeof(g_err), "bad argument: brl_mlp_gemm_x3: epilogue / k a multiple of 32 (oracle shim)");
    return BRL_E_ARG;
  }
  return brl_mlp_gemm(device, layout, epilogue, a, lda, b, ldb, c, ldc, m, n, k, act, bias, gate, ldg, colsum, NULL, s);
}
/* the plane split and the layer on planes, restated plainly: hi / mid / lo by truncation; the product from the planes' SUMS in float64 */
static float shim_bf(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static void shim_split1(float x, uint16_t out[3]) {
  uint32_t u; memcpy(&u, &x, 4);
  uint32_t uh = u & 0xffff0000u; float h; memcpy(&h, &uh, 4);
  float r = x - h; uint32_t ur; memcpy(&ur, &r, 4);
  uint32_t um = ur & 0xffff0000u; float md; memcpy(&md, &um, 4);
  float l = r - md; uint32_t ul; memcpy(&ul, &l, 4);
  out[0] = (uint16_t)(u >> 16); out[1] = (uint16_t)(ur >> 16); out[2] = (uint16_t)(ul >> 16);
}
int brl_split_planes(int device, const float *x, int64_t n, uint16_t *planes, int64_t plane_stride, void *s) {
  (void)device; (void)s;
  if (!x || !planes || n <= 0 || n % 4 || plane_stride < n) {
    snprintf(g_err, sizeof(g_err), "bad argument: brl_split_planes (oracle shim)");
    return BRL_E_ARG;
  }
  for (int64_t i = 0; i < n; i++) {
    uint16_t pl[3];
    shim_split1(x[i], pl);
    planes[i] = pl[0]; planes[plane_stride + i] = pl[1]; planes[2 * plane_stride + i] = pl[2];
  }
  return BRL_OK;
}
int brl_linear_x3p(int device, const uint16_t *xp, int npx, int64_t ldx, int64_t sx, const uint16_t *wp, int64_t ldw, int64_t sw,
                   const float *bias, int relu, float *y, int64_t ldy, uint16_t *yp, int64_t ldyp, int64_t syp, int64_t m, int64_t n,
                   int64_t k, void *s) {
  (void)device; (void)s;
  if (!xp || !wp || !bias || (!y && !yp) || (npx != 1 && npx != 3) || n % 128 || k % 32) {
    snprintf(g_err, sizeof(g_err), "bad argument: brl_linear_x3p (oracle shim)");
    return BRL_E_ARG;
  }
  for (int64_t i = 0; i < m; i++)
    for (int64_t j = 0; j < n; j++) {
      double acc = 0.0;
      for (int64_t q = 0; q < k; q++) {
        double xv = 0.0, wv = 0.0;
        for (int p = 0; p < npx; p++) xv += (double)shim_bf(xp[p * sx + i * ldx + q]);
        for (int p = 0; p < 3; p++) wv += (double)shim_bf(wp[p * sw + j * ldw + q]);
        acc += xv * wv;
      }
      float o = (float)(acc + (double)bias[j]);
      if (relu && o < 0.0f) o = 0.0f;
      if (y) y[i * ldy + j] = o;
      if (yp) {
        uint16_t pl[3];
        shim_split1(o, pl);
        for (int p = 0; p < 3; p++) yp[p * syp + i * ldyp + j] = pl[p];
      }
    }
  return BRL_OK;
}
int brl_mlp_gemm_x3_group(int device, int layout, int count, const float *const *a, const int64_t *lda, const float *const *b,
                          const int64_t *ldb, float *const *c, const int64_t *ldc, const int64_t *m, const int64_t *n,
                          const int64_t *k, void *s) {   /* the plain definition, product by product */
  if (count < 1 || count > 8) {
    snprintf(g_err, sizeof(g_err), "bad argument: brl_mlp_gemm_x3_group: count (oracle shim)");
    return BRL_E_ARG;
  }
  for (int i = 0; i < count; i++)
    if (!k || k[i] % 32) {
      snprintf(g_err, sizeof(g_err), "bad argument: brl_mlp_gemm_x3_group: k a multiple of 32 (oracle shim)");
      return BRL_E_ARG;
    }
  return brl_mlp_gemm_group(device, layout, count, a, lda, b, ldb, c, ldc, m, n, k, s);
}
/* the policy network's forward for selected rows, float64 accumulation (the checker of brl_mlp_forward_rows: src/models.py:23-33) */
int brl_mlp_forward_rows(int device, const brl_mlp_ref *net, const uint8_t *obs, const int64_t *rows, int64_t m, float *scratch,
                         int64_t scratch_len, float *out, int64_t ldo, void *s) {
  (void)device; (void)s;
  if (!net || !obs || !scratch || !out || m <= 0 || net->nlayers < 1 || net->nlayers > 8 || net->in_features != BRL_OBS_SIZE ||
      net->hidden <= 0 || net->hidden % 4 || net->hidden > 1024 || (net->act != 0 && net->act != 1) || ldo < BRL_NUM_ACTIONS + 1 ||
      scratch_len < m * (BRL_OBS_SIZE + 2 * net->hidden)) {
    snprintf(g_err, sizeof(g_err), "bad argument: brl_mlp_forward_rows (oracle shim)");
    return BRL_E_ARG;
  }
  const int64_t H = net->hidden;
  float *x = scratch, *buf[2] = {scratch + m * BRL_OBS_SIZE, scratch + m * BRL_OBS_SIZE + m * H};
  for (int64_t r = 0; r < m; r++)
    for (int q = 0; q < BRL_OBS_SIZE; q++) x[r * BRL_OBS_SIZE + q] = (float)obs[(rows ? rows[r] : r) * BRL_OBS_SIZE + q];
  const float *cur = x;
  int64_t k = BRL_OBS_SIZE;
  for (int l = 0; l < net->nlayers; l++) {
    float *dst = buf[l & 1];
    for (int64_t r = 0; r < m; r++)
      for (int64_t j = 0; j < H; j++) {
        double acc = net->b[l][j];
        for (int64_t q = 0; q < k; q++) acc += (double)cur[r * k + q] * (double)net->w[l][j * k + q];
        dst[r * H + j] = (float)(net->act == 0 ? (acc > 0 ? acc : 0) : tanh(acc));
      }
    cur = dst;
    k = H;
  }
  for (int64_t r = 0; r < m; r++)
    for (int hd = 0; hd < BRL_NUM_ACTIONS + 1; hd++) {
      const float *w = hd < BRL_NUM_ACTIONS ? net->actor_w + (int64_t)hd * H : net->critic_w;
      double acc = hd < BRL_NUM_ACTIONS ? net->actor_b[hd] : net->critic_b[0];
      for (int64_t q = 0; q < H; q++) acc += (double)cur[r * H + q] * (double)w[q];
      out[(rows ? rows[r] : r) * ldo + hd] = (float)acc;
    }
  return BRL_OK;
}
int brl_mlp_gemm_dh_heads_dw(int device, const float *dz, int64_t lddz, const float *w, int64_t ldw, float *out, int64_t ldo,
                             int64_t m, int64_t n, int64_t k, int act, const float *gate, int64_t ldg, float *colsum,
                             const float *dheads, const float *h, int64_t ldh, int64_t batch, int64_t hidden, int nsplit,
                             float *dw_partials, float *db_partials, const float *loss_partials, const float *gram_partials,
                             int64_t ngroups, const int32_t *row_index, float *stat_sums, float *gram_sums, void *s) {
  (void)device; (void)dz; (void)lddz; (void)w; (void)ldw; (void)out; (void)ldo; (void)m; (void)n; (void)k; (void)act; (void)gate; (void)ldg;
  (void)colsum; (void)dheads; (void)h; (void)ldh; (void)batch; (void)hidden; (void)nsplit; (void)dw_partials; (void)db_partials;
  (void)loss_partials; (void)gram_partials; (void)ngroups; (void)row_index; (void)stat_sums; (void)gram_sums; (void)s;
  NOT_HERE("brl_mlp_gemm_dh_heads_dw");
}
/* clip + Adam on rank slices of the bucketed flat buffers (include/brl_hip.h: brl_shard_geom), host pointers.  The partial sums
 * are per (rank, bucket, sub-block) exactly as the device kernels lay them out (float accumulation in index order inside a
 * sub-block: not bit-equal to the device's tree order, the same to ~1e-6); the sweep is torch.optim.Adam's arithmetic in float. */
static int shard_ok(const brl_shard_geom *G, int lo, int hi) {
  if (!G || G->nbuckets < 1 || G->nbuckets > 12 || G->world < 1 || G->nsub < 1 || lo < 0 || lo >= hi || hi > G->world) return 0;
  for (int b = 0; b < G->nbuckets; b++)
    if (G->off[b] < 0 || G->off[b] % 4 || G->len[b] <= 0 || G->len[b] % 4) return 0;
  return 1;
}
int brl_adam_shard_norm(int device, const float *g, const brl_shard_geom *G, int rank_lo, int rank_hi, float grad_scale, float *partials,
                        float *step, int32_t *mb_index, void *s) {
  (void)device; (void)s;
  if (!g || !partials || !step || !shard_ok(G, rank_lo, rank_hi)) return fail("bad argument: brl_adam_shard_norm (oracle shim)");
  for (int r = rank_lo; r < rank_hi; r++)
    for (int b = 0; b < G->nbuckets; b++) {
      const int64_t len4 = G->len[b] / 4, chunk4 = (len4 + G->nsub - 1) / G->nsub, base = G->off[b] + (int64_t)r * G->len[b];
      for (int j = 0; j < G->nsub; j++) {
        int64_t lo = (int64_t)j * chunk4 * 4, hi = lo + chunk4 * 4;
        if (hi > G->len[b]) hi = G->len[b];
        float acc = 0.0f;
        for (int64_t i = lo; i < hi; i++) {
          const float x = g[base + i] * grad_scale;
          acc += x * x;
        }
        partials[((int64_t)r * G->nbuckets + b) * G->nsub + j] = acc;
      }
    }
  *step += 1.0f;
  if (mb_index) *mb_index += 1;
  return BRL_OK;
}
int brl_adam_shard_apply(int device, float *p, const float *g, float *m, float *v, const brl_shard_geom *G, int rank_lo, int rank_hi,
                         const float *partials, const float *step, float lr, const float *lr_dev, float beta1, float beta2, float eps,
                         float max_norm, float grad_scale, float *norm_out, const void *gather_args, int64_t mbs, void *s) {
  (void)device; (void)s; (void)mbs;
  if (!p || !g || !m || !v || !partials || !step || !shard_ok(G, rank_lo, rank_hi)) return fail("bad argument: brl_adam_shard_apply (oracle shim)");
  if (gather_args) NOT_HERE("brl_adam_shard_apply with the minibatch gather");
  float sum = 0.0f;
  for (int64_t i = 0; i < (int64_t)G->world * G->nbuckets * G->nsub; i++) sum += partials[i];
  const float norm = sqrtf(sum);
  const float coef = max_norm > 0.0f ? fminf(max_norm / (norm + 1e-6f), 1.0f) : 1.0f;
  if (norm_out) *norm_out = norm;
  const float scale = coef * grad_scale, t = *step, rate = lr_dev ? *lr_dev : lr;
  const float bc1 = 1.0f - powf(beta1, t), bc2 = 1.0f - powf(beta2, t), step_size = rate / bc1, bc2_sqrt = sqrtf(bc2);
  for (int r = rank_lo; r < rank_hi; r++)
    for (int b = 0; b < G->nbuckets; b++) {
      const int64_t base = G->off[b] + (int64_t)r * G->len[b];
      for (int64_t i = base; i < base + G->len[b]; i++) {
        const float gs = g[i] * scale;
        m[i] = m[i] + (gs - m[i]) * (1.0f - beta1);
        v[i] = v[i] * beta2 + gs * gs * (1.0f - beta2);
        p[i] -= step_size * (m[i] / (sqrtf(v[i]) / bc2_sqrt + eps));
      }
    }
  return BRL_OK;
}
int brl_adam_clip_fin_gather(int device, float *p, float *g, float *m, float *v, int64_t n, float *step, float lr,
                             const float *lr_dev, float beta1, float beta2, float eps, float max_norm, float *scratch,
                             int64_t scratch_len, int32_t *mb_index, float *norm_out, const void *gather_args, int64_t mbs,
                             int nseg, const float *const *partials, const int64_t *cols, const int64_t *tiles, float *const *out,
                             void *s) {
  (void)device; (void)p; (void)g; (void)m; (void)v; (void)n; (void)step; (void)lr; (void)lr_dev; (void)beta1; (void)beta2; (void)eps;
  (void)max_norm; (void)scratch; (void)scratch_len; (void)mb_index; (void)norm_out; (void)gather_args; (void)mbs; (void)nseg;
  (void)partials; (void)cols; (void)tiles; (void)out; (void)s;
  NOT_HERE("brl_adam_clip_fin_gather");
}
int brl_act_bwd_colsum_heads_dw(int device, float *dz, const float *hh, int64_t rows, int64_t cols, int64_t ld, int act,
                                float *scratch, const float *dheads, const float *h, int64_t ldh, int64_t batch, int64_t hidden,
                                int nsplit, float *dw_partials, float *db_partials, const float *loss_partials,
                                const float *gram_partials, int64_t ngroups, const int32_t *row_index, float *stat_sums,
                                float *gram_sums, void *s) {
  (void)device; (void)dz; (void)hh; (void)rows; (void)cols; (void)ld; (void)act; (void)scratch; (void)dheads; (void)h; (void)ldh;
  (void)batch; (void)hidden; (void)nsplit; (void)dw_partials; (void)db_partials; (void)loss_partials; (void)gram_partials;
  (void)ngroups; (void)row_index; (void)stat_sums; (void)gram_sums; (void)s;
  NOT_HERE("brl_act_bwd_colsum_heads_dw");
}
int brl_eval_step(brl_handle *h, const uint64_t *si, uint64_t *so, int64_t n, const float *l1, int64_t s1, const float *l2,
                  int64_t s2, const brl_table_info *ta, const brl_table_info *tb, const brl_eval_stats *st, int bs,
                  float *cr, float *rs, int32_t *ao, uint8_t *o, uint8_t *m, float *r, uint8_t *t, int32_t *c, void *s) {
  (void)h; (void)si; (void)so; (void)n; (void)l1; (void)s1; (void)l2; (void)s2; (void)ta; (void)tb; (void)st; (void)bs; (void)cr; (void)rs; (void)ao; (void)o; (void)m; (void)r; (void)t; (void)c; (void)s;
  NOT_HERE("brl_eval_step");
}
int brl_eval_reduce(brl_handle *h, int64_t n, const brl_table_info *ta, const brl_table_info *tb, const int32_t *bc,
                    const uint64_t *st, int64_t *out, void *s) {
  (void)h; (void)n; (void)ta; (void)tb; (void)bc; (void)st; (void)out; (void)s;
  NOT_HERE("brl_eval_reduce");
}
int brl_ppo_loss(int device, const float *lg, int64_t ls, const float *v, const uint8_t *m, const int32_t *a, const float *ov,
                 const float *olp, const float *g, const float *t, int64_t b, float ce, float vc, float ec, int mk, int vcl,
                 float *dl, float *dv, float *pt, float *ip, void *s) {
  (void)device; (void)lg; (void)ls; (void)v; (void)m; (void)a; (void)ov; (void)olp; (void)g; (void)t; (void)b; (void)ce; (void)vc; (void)ec; (void)mk; (void)vcl; (void)dl; (void)dv; (void)pt; (void)ip; (void)s;
  NOT_HERE("brl_ppo_loss");
}
int brl_ppo_stats(int device, const float *pt, int64_t b, const float *gram, float vc, float ec, float *out, void *s) {
  (void)device; (void)pt; (void)b; (void)gram; (void)vc; (void)ec; (void)out; (void)s;
  NOT_HERE("brl_ppo_stats");
}
/* the fused PPO minibatch step (GEMM-side helpers of brl_amd/update.py::FusedMinibatch): GPU library only; the float64
 * restatement the update tests compare with is tests/ppo_numpy.py */
int brl_policy_step_ex(brl_handle *h, const uint64_t *si, uint64_t *so, int64_t n, const float *lg, int64_t ls, int mode,
                       const uint32_t *db, uint32_t dof, int ar, int32_t *a, float *lp, uint8_t *obs, uint8_t *m, float *ra,
                       uint8_t *ta, int32_t *cp, const brl_macro_ext *ext, void *s) {
  (void)h; (void)si; (void)so; (void)n; (void)lg; (void)ls; (void)mode; (void)db; (void)dof; (void)ar; (void)a; (void)lp; (void)obs; (void)m; (void)ra; (void)ta; (void)cp; (void)ext; (void)s;
  NOT_HERE("brl_policy_step_ex");
}
int brl_eval_step_team(brl_handle *h, const uint64_t *si, uint64_t *so, int64_t n, const float *lg, int64_t st, int team,
                       const brl_table_info *ta, const brl_table_info *tb, const brl_eval_stats *es, int bs, float *cr, float *rs,
                       int32_t *ao, uint8_t *obs, uint8_t *m, float *rw, uint8_t *tm, int32_t *cp, float *of, void *s) {
  (void)h; (void)si; (void)so; (void)n; (void)lg; (void)st; (void)team; (void)ta; (void)tb; (void)es; (void)bs; (void)cr; (void)rs; (void)ao; (void)obs; (void)m; (void)rw; (void)tm; (void)cp; (void)of; (void)s;
  NOT_HERE("brl_eval_step_team");
}
int brl_rollout_random_gae(brl_handle *h, uint64_t *state, int64_t n, int T, uint32_t draw_base, float reward_scale,
                           const brl_transition *out, uint8_t *last_obs, uint8_t *last_mask, int64_t *tc, const float *last_val,
                           float gamma, float gl, float *adv, float *tgt, void *s) {
  int rc = brl_rollout_random(h, state, n, T, 1, draw_base, reward_scale, out, last_obs, last_mask, tc, s);
  if (rc != 0) return rc;
  return brl_gae(h, out->done, out->value, out->reward, last_val, gamma, gl, T, n, adv, tgt, s);
}
/* the head kernels of the fused PPO minibatch step (brl_amd/csrc/ppo_heads.hpp): GPU library only, like the entries above */
int brl_ppo_heads_loss_split(int device, const float *h, int64_t ldh, const float *hw, const float *hb, int64_t hidden, const uint8_t *m,
                       const int32_t *a, const float *ov, const float *olp, const float *g, const float *t, int64_t b, float ce,
                       float vc, float ec, int mk, int vcl, int rs, float *ho, float *dh, float *pt, float *gp, float *hp, int ks, void *s) {
  (void)device; (void)h; (void)ldh; (void)hw; (void)hb; (void)hidden; (void)m; (void)a; (void)ov; (void)olp; (void)g; (void)t; (void)b; (void)ce; (void)vc; (void)ec; (void)mk; (void)vcl; (void)rs; (void)ho; (void)dh; (void)pt; (void)gp; (void)hp; (void)ks; (void)s;
  NOT_HERE("brl_ppo_heads_loss_split");
}
int brl_ppo_heads_bwd(int device, const float *dheads, const float *h, int64_t ldh, const float *hw, int64_t b, int64_t hidden,
                      int act, int nsplit, float *dwp, float *dbp, float *dh, float *ts, const float *lp, const float *gp, int64_t ng,
                      const int32_t *ri, float *ss, float *gs, void *s) {
  (void)lp; (void)gp; (void)ng; (void)ri; (void)ss; (void)gs; (void)device; (void)dheads; (void)h; (void)ldh; (void)hw; (void)b; (void)hidden; (void)act; (void)nsplit; (void)dwp; (void)dbp; (void)dh; (void)ts; (void)s;
  NOT_HERE("brl_ppo_heads_bwd");
}
int brl_ppo_stats_gram(int device, const float *pt, int64_t np, int64_t b, const float *gp, int64_t ng, float vc, float ec, float ic,
                       float *out, const int32_t *ri, float *vec, void *s) {
  (void)ic; (void)device; (void)pt; (void)np; (void)b; (void)gp; (void)ng; (void)vc; (void)ec; (void)out; (void)ri; (void)vec; (void)s;
  NOT_HERE("brl_ppo_stats_gram");
}
int brl_act_bwd_colsum(int device, float *dh, const float *h, int64_t rows, int64_t cols, int64_t ld, int act, float *scr, void *s) {
  (void)device; (void)dh; (void)h; (void)rows; (void)cols; (void)ld; (void)act; (void)scr; (void)s;
  NOT_HERE("brl_act_bwd_colsum");
}
int brl_bias_finalize_ex(int device, int nseg, const float *const *parts, const int64_t *cols, const int64_t *tiles,
                         float *const *out, void *s) {
  (void)device; (void)nseg; (void)parts; (void)cols; (void)tiles; (void)out; (void)s;
  NOT_HERE("brl_bias_finalize_ex");
}
int brl_mlp_gemm_group(int device, int layout, int count, const float *const *a, const int64_t *lda, const float *const *b,
                       const int64_t *ldb, float *const *c, const int64_t *ldc, const int64_t *m, const int64_t *n,
                       const int64_t *k, void *s) {   /* the plain definition, product by product (brl_mlp_gemm above) */
  for (int i = 0; i < count; i++) {
    const int rc = brl_mlp_gemm(device, layout, BRL_GEMM_EPI_NONE, a[i], lda[i], b[i], ldb[i], c[i], ldc[i], m[i], n[i], k[i], 0, NULL,
                                NULL, 0, NULL, NULL, s);
    if (rc) return rc;
  }
  return BRL_OK;
}
/* brl_fair_forward on the host: src/models.py:34-69 written out, float64 accumulation (act: 0 ReLU, 1 tanh) */
static void fair_lin(const float *w, const float *b, const double *x, int in, int out, double *y) {
  for (int o = 0; o < out; o++) {
    double a = b[o];
    for (int i = 0; i < in; i++) a += (double)w[(int64_t)o * in + i] * x[i];
    y[o] = a;
  }
}
int brl_fair_forward(int device, const brl_fair_net *net, const float *x, int64_t rows, int act, float *logits, float *value, void *s) {
  (void)device; (void)s;
  if (!net || !x || !logits || !value || rows < 0 || (act != 0 && act != 1)) {
    snprintf(g_err, sizeof(g_err), "bad argument: brl_fair_forward (oracle shim)");
    return BRL_E_ARG;
  }
#define FAIR_ACT(v) (act == 0 ? ((v) > 0.0 ? (v) : 0.0) : tanh(v))
  for (int64_t r = 0; r < rows; r++) {
    double in0[480], cat[680], a[200], bb[200], sc[200], h[39];
    for (int i = 0; i < 480; i++) in0[i] = x[r * 480 + i];
    fair_lin(net->w[0], net->b[0], in0, 480, 200, sc);                      /* shortcut_1 = the pre-activation */
    for (int i = 0; i < 200; i++) a[i] = FAIR_ACT(sc[i]);
    for (int blk = 0; blk < 2; blk++) {                                     /* two residual blocks: layers 1-2, 3-4 */
      fair_lin(net->w[1 + 2 * blk], net->b[1 + 2 * blk], a, 200, 200, bb);
      for (int i = 0; i < 200; i++) bb[i] = FAIR_ACT(bb[i]);
      fair_lin(net->w[2 + 2 * blk], net->b[2 + 2 * blk], bb, 200, 200, a);
      for (int i = 0; i < 200; i++) { sc[i] = FAIR_ACT(a[i]) + sc[i]; a[i] = FAIR_ACT(sc[i]); }
    }
    fair_lin(net->w[5], net->b[5], sc, 200, 200, cat);                      /* (no activation) */
    for (int i = 0; i < 480; i++) cat[200 + i] = in0[i];
    fair_lin(net->w[6], net->b[6], cat, 680, 200, sc);
    for (int i = 0; i < 200; i++) a[i] = FAIR_ACT(sc[i]);
    for (int blk = 0; blk < 2; blk++) {                                     /* layers 7-8, 9-10 */
      fair_lin(net->w[7 + 2 * blk], net->b[7 + 2 * blk], a, 200, 200, bb);
      for (int i = 0; i < 200; i++) bb[i] = FAIR_ACT(bb[i]);
      fair_lin(net->w[8 + 2 * blk], net->b[8 + 2 * blk], bb, 200, 200, a);
      for (int i = 0; i < 200; i++) { sc[i] = FAIR_ACT(a[i]) + sc[i]; a[i] = FAIR_ACT(sc[i]); }
    }
    fair_lin(net->head_w, net->head_b, sc, 200, 39, h);
    for (int i = 0; i < 38; i++) logits[r * 38 + i] = (float)h[i];
    value[r] = (float)h[38];
  }
#undef FAIR_ACT
  return BRL_OK;
}
int brl_bias_finalize_rows(int device, int nseg, const float *const *parts, const int64_t *cols, const int64_t *tiles,
                           float *const *out, int first_row_seg, const int32_t *row_index, void *s) {
  (void)device; (void)nseg; (void)parts; (void)cols; (void)tiles; (void)out; (void)first_row_seg; (void)row_index; (void)s;
  NOT_HERE("brl_bias_finalize_rows");
}
int brl_fair_chain(int device, const brl_fair_net *net, const float *x0, const uint8_t *mask, const int32_t *action,
                   const float *old_value, const float *old_log_prob, const float *gae, const float *targets, int64_t batch,
                   float clip_eps, float vf_coef, float ent_coef, int masked, int value_clipping, int reward_scaling, int act,
                   const brl_fair_work *work, void *s) {
  (void)device; (void)net; (void)x0; (void)mask; (void)action; (void)old_value; (void)old_log_prob; (void)gae; (void)targets;
  (void)batch; (void)clip_eps; (void)vf_coef; (void)ent_coef; (void)masked; (void)value_clipping; (void)reward_scaling; (void)act;
  (void)work; (void)s;
  NOT_HERE("brl_fair_chain");
}
int brl_ppo_stats_rows(int device, const float *ss, const float *gs, int64_t rows, int64_t b, float vc, float ec, float ic, float *out,
                       void *s) {
  (void)ic; (void)device; (void)ss; (void)gs; (void)rows; (void)b; (void)vc; (void)ec; (void)out; (void)s;
  NOT_HERE("brl_ppo_stats_rows");
}
int brl_mb_gather_bind(int device, const brl_transition *flat, const float *adv, const float *tg, const int64_t *perm, const int32_t *mbi,
                       int64_t mbs, float *x0, uint8_t *m, int32_t *a, float *ov, float *olp, float *go, float *to, int64_t ns, void *ad,
                       void *s) {
  (void)ns; (void)device; (void)flat; (void)adv; (void)tg; (void)perm; (void)mbi; (void)mbs; (void)x0; (void)m; (void)a; (void)ov; (void)olp; (void)go; (void)to; (void)ad; (void)s;
  NOT_HERE("brl_mb_gather_bind");
}
int brl_mb_gather_dev(int device, const void *ad, int64_t mbs, void *s) {
  (void)device; (void)ad; (void)mbs; (void)s;
  NOT_HERE("brl_mb_gather_dev");
}
int brl_ppo_illegal_grad(int device, const float *hd, const uint8_t *m, const float *vec, float ic, int64_t b, float *dh, void *s) {
  (void)device; (void)hd; (void)m; (void)vec; (void)ic; (void)b; (void)dh; (void)s;
  NOT_HERE("brl_ppo_illegal_grad");
}
