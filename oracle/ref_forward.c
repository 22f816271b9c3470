/*
 * oracle/ref_forward.c -- CPU restatement of the FastLLM decoder forward pass.
 * TEST INFRASTRUCTURE ONLY (see ref_forward.h for scope and parity status).
 *
 * What is restated, and from where:
 *   - wrapper semantics (config defaults, validation, call-counter offset, generate
 *     loop): the reference's own Rust sources, cited at each function;
 *   - arithmetic: candle-transformers 0.8.x models::{llama,mistral,qwen2}, which is NOT in
 *     /root/reference (Cargo.toml:19-21).  Restated from the published architectures and
 *     the numeric choices listed in SURVEY.md Appendix A ([UPSTREAM-RECALLED]).
 *
 * Plain C11 + OpenMP.  fp32 arithmetic, fp32 accumulation in 16 independent partial
 * sums (fixed order, so results do not depend on the thread count).
 */
#include "ref_forward.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static _Thread_local char g_err[512];
const char *orc_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return -1; } while (0)

/* ---------------------------------------------------------------- scalars */
static inline float bf16_to_f32(uint16_t b) {
    uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f;
}
static inline float round_bf16f(float f) {            /* RNE to bf16, NaN kept */
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return f;
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    memcpy(&f, &u, 4); return f;
}
static inline float f16_to_f32(uint16_t h) {
    uint32_t s = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, m = h & 0x3ff, u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int sh = 0; while (!(m & 0x400)) { m <<= 1; sh++; } m &= 0x3ff; u = s | ((uint32_t)(113 - sh) << 23) | (m << 13); }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}

/* ---------------------------------------------------------------- model */
typedef struct {
    const float *wf;      /* fp32 weights [N][K] or NULL */
    const uint16_t *wb;   /* bf16 weights [N][K] or NULL (kept as stored: exact) */
    const float *bias;    /* [N] or NULL */
    int64_t N, K;
} linear_t;

typedef struct {
    linear_t q, k, v, o, gate, up, down;
    const float *ln1, *ln2;
} layer_t;

struct orc_model {
    orc_config cfg;
    int64_t h, inter, V, L, H, Hkv, d, window;   /* window < 0: none */
    float eps, scale;
    float *inv_freq;      /* [d/2] */
    int round_bf16;
    int nthreads;
    layer_t *layers;
    const float *embed_f; const uint16_t *embed_b;
    const float *norm;
    linear_t lm_head;
    void **owned; size_t n_owned;
    orc_allreduce_fn allreduce; void *allreduce_ctx;
};

struct orc_cache {
    const orc_model *m;
    size_t max_seq, len;
    float *k, *v;         /* [L][Hkv][max_seq][d] */
};

static void *own(orc_model *m, void *p) {
    m->owned = (void **)realloc(m->owned, (m->n_owned + 1) * sizeof(void *));
    m->owned[m->n_owned++] = p; return p;
}

static const orc_tensor *find_tensor(const orc_tensor *t, size_t n, const char *name) {
    for (size_t i = 0; i < n; i++) if (strcmp(t[i].name, name) == 0) return &t[i];
    return NULL;
}

/* VarBuilder::from_tensors(tensors, dtype, device) (llama.rs:112, mistral.rs:190,
 * qwen.rs:108): look the tensor up by HF name and cast it to the compute dtype.
 * bf16 sources are kept as bf16 (exactly representable in fp32; halves host memory). */
static int get_matrix(orc_model *m, const orc_tensor *ts, size_t n, const char *name,
                      int64_t N, int64_t K, const float **wf, const uint16_t **wb) {
    const orc_tensor *t = find_tensor(ts, n, name);
    if (!t) FAIL("missing tensor %s", name);
    if (t->ndim != 2 || t->shape[0] != N || t->shape[1] != K)
        FAIL("shape mismatch for %s: got [%lld,%lld] want [%lld,%lld]", name,
             (long long)t->shape[0], (long long)(t->ndim > 1 ? t->shape[1] : 0), (long long)N, (long long)K);
    size_t cnt = (size_t)N * (size_t)K;
    *wf = NULL; *wb = NULL;
    if (t->dtype == ORC_BF16) {
        uint16_t *p = (uint16_t *)own(m, malloc(cnt * 2));
        if (!p) FAIL("out of memory");
        memcpy(p, t->data, cnt * 2); *wb = p;
    } else if (t->dtype == ORC_F32) {
        float *p = (float *)own(m, malloc(cnt * 4));
        if (!p) FAIL("out of memory");
        memcpy(p, t->data, cnt * 4); *wf = p;
    } else if (t->dtype == ORC_F16) {
        float *p = (float *)own(m, malloc(cnt * 4));
        if (!p) FAIL("out of memory");
        const uint16_t *s = (const uint16_t *)t->data;
        for (size_t i = 0; i < cnt; i++) p[i] = f16_to_f32(s[i]);
        *wf = p;
    } else FAIL("unsupported dtype %d for %s", t->dtype, name);
    return 0;
}

static int get_vector(orc_model *m, const orc_tensor *ts, size_t n, const char *name,
                      int64_t N, const float **out) {
    const orc_tensor *t = find_tensor(ts, n, name);
    if (!t) FAIL("missing tensor %s", name);
    if (t->ndim != 1 || t->shape[0] != N) FAIL("shape mismatch for %s", name);
    float *p = (float *)own(m, malloc((size_t)N * 4));
    if (!p) FAIL("out of memory");
    if (t->dtype == ORC_F32) memcpy(p, t->data, (size_t)N * 4);
    else if (t->dtype == ORC_BF16) for (int64_t i = 0; i < N; i++) p[i] = bf16_to_f32(((const uint16_t *)t->data)[i]);
    else if (t->dtype == ORC_F16) for (int64_t i = 0; i < N; i++) p[i] = f16_to_f32(((const uint16_t *)t->data)[i]);
    else FAIL("unsupported dtype for %s", name);
    *out = p; return 0;
}

static int get_linear(orc_model *m, const orc_tensor *ts, size_t n, const char *prefix,
                      int64_t N, int64_t K, int bias, linear_t *lin) {
    char name[256];
    snprintf(name, sizeof name, "%s.weight", prefix);
    lin->N = N; lin->K = K; lin->bias = NULL;
    if (get_matrix(m, ts, n, name, N, K, &lin->wf, &lin->wb)) return -1;
    if (bias) {
        snprintf(name, sizeof name, "%s.bias", prefix);
        if (get_vector(m, ts, n, name, N, &lin->bias)) return -1;
    }
    return 0;
}

int orc_model_create(const orc_config *cfg, const orc_tensor *ts, size_t n, int round_bf16,
                     orc_model **out) {
    if (!cfg || !out) FAIL("null argument");
    orc_model *m = (orc_model *)calloc(1, sizeof *m);
    if (!m) FAIL("out of memory");
    m->cfg = *cfg;
    m->h = cfg->hidden_size; m->inter = cfg->intermediate_size; m->V = cfg->vocab_size;
    m->L = cfg->num_hidden_layers; m->H = cfg->num_attention_heads;
    /* num_key_value_heads.unwrap_or(num_attention_heads): llama.rs:39, mistral.rs:97, qwen.rs:45 */
    m->Hkv = cfg->num_key_value_heads > 0 ? cfg->num_key_value_heads : m->H;
    if (m->h <= 0 || m->inter <= 0 || m->V <= 0 || m->L <= 0 || m->H <= 0) { free(m); FAIL("bad config: non-positive dimension"); }
    if (cfg->head_dim > 0) m->d = cfg->head_dim;
    else {
        /* config.rs:32-44 / mistral.rs:67-76: hidden_size % heads == 0, head_dim even */
        m->d = m->h / m->H;
        if (m->d * m->H != m->h) { free(m); FAIL("bad config: hidden_size must be divisible by num_attention_heads"); }
    }
    if (m->d % 2) { free(m); FAIL("bad config: head_dim must be even for RoPE embeddings"); }
    /* config.rs:46-54 / mistral.rs:109-112 */
    if (m->H % m->Hkv) { free(m); FAIL("bad config: num_attention_heads must be divisible by num_key_value_heads"); }
    m->eps = (float)cfg->rms_norm_eps;
    /* rope_theta.unwrap_or(10000.0): llama.rs:41, mistral.rs:137, qwen.rs:47 */
    float theta = (float)(cfg->rope_theta > 0 ? cfg->rope_theta : 10000.0);
    /* sliding_window: Some(cfg or 4096) mistral.rs:139; qwen.rs:49; Llama has none */
    if (cfg->family == ORC_LLAMA) m->window = -1;
    else m->window = cfg->sliding_window > 0 ? cfg->sliding_window : (cfg->sliding_window < 0 ? -1 : 4096);
    m->scale = (float)(1.0 / sqrt((double)m->d));        /* App. A.3 */
    m->round_bf16 = round_bf16;
    m->nthreads = 0;
    m->inv_freq = (float *)own(m, malloc((size_t)(m->d / 2) * 4));
    for (int64_t j = 0; j < m->d / 2; j++)               /* App. A.4: 1 / theta^(2j/d) in f32 */
        m->inv_freq[j] = 1.0f / powf(theta, (float)(2 * j) / (float)m->d);

    int rc = 0; char p[200];
    const orc_tensor *emb = find_tensor(ts, n, "model.embed_tokens.weight");
    if (!emb) { snprintf(g_err, sizeof g_err, "missing tensor model.embed_tokens.weight"); rc = -1; }
    if (!rc) rc = get_matrix(m, ts, n, "model.embed_tokens.weight", m->V, m->h, &m->embed_f, &m->embed_b);
    m->layers = (layer_t *)own(m, calloc((size_t)m->L, sizeof(layer_t)));
    int bias = cfg->qkv_bias != 0;
    for (int64_t l = 0; l < m->L && !rc; l++) {
        layer_t *ly = &m->layers[l];
#define LIN(field, nm, N_, K_, b_) do { snprintf(p, sizeof p, "model.layers.%lld." nm, (long long)l); \
        if (!rc) rc = get_linear(m, ts, n, p, (N_), (K_), (b_), &ly->field); } while (0)
        LIN(q, "self_attn.q_proj", m->H * m->d, m->h, bias);
        LIN(k, "self_attn.k_proj", m->Hkv * m->d, m->h, bias);
        LIN(v, "self_attn.v_proj", m->Hkv * m->d, m->h, bias);
        LIN(o, "self_attn.o_proj", m->h, m->H * m->d, 0);
        LIN(gate, "mlp.gate_proj", m->inter, m->h, 0);
        LIN(up, "mlp.up_proj", m->inter, m->h, 0);
        LIN(down, "mlp.down_proj", m->h, m->inter, 0);
#undef LIN
        snprintf(p, sizeof p, "model.layers.%lld.input_layernorm.weight", (long long)l);
        if (!rc) rc = get_vector(m, ts, n, p, m->h, &ly->ln1);
        snprintf(p, sizeof p, "model.layers.%lld.post_attention_layernorm.weight", (long long)l);
        if (!rc) rc = get_vector(m, ts, n, p, m->h, &ly->ln2);
    }
    if (!rc) rc = get_vector(m, ts, n, "model.norm.weight", m->h, &m->norm);
    if (!rc) {
        if (find_tensor(ts, n, "lm_head.weight")) rc = get_linear(m, ts, n, "lm_head", m->V, m->h, 0, &m->lm_head);
        else if (cfg->family == ORC_QWEN2) {   /* candle qwen2 falls back to the embedding (App. A.1) */
            m->lm_head.wf = m->embed_f; m->lm_head.wb = m->embed_b; m->lm_head.N = m->V; m->lm_head.K = m->h; m->lm_head.bias = NULL;
        } else { snprintf(g_err, sizeof g_err, "missing tensor lm_head.weight"); rc = -1; }
    }
    if (rc) { orc_model_destroy(m); return -1; }
    *out = m; return 0;
}

void orc_model_destroy(orc_model *m) {
    if (!m) return;
    for (size_t i = 0; i < m->n_owned; i++) free(m->owned[i]);
    free(m->owned); free(m);
}
void orc_model_set_allreduce(orc_model *m, orc_allreduce_fn fn, void *ctx) { m->allreduce = fn; m->allreduce_ctx = ctx; }
void orc_model_set_threads(orc_model *m, int n) { m->nthreads = n; }
int orc_model_threads(const orc_model *m) {
    if (m->nthreads > 0) return m->nthreads;
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------- cache */
int orc_cache_create(const orc_model *m, size_t max_seq, orc_cache **out) {
    if (!m || !out || max_seq == 0) FAIL("bad argument");
    orc_cache *c = (orc_cache *)calloc(1, sizeof *c);
    if (!c) FAIL("out of memory");
    size_t cnt = (size_t)m->L * (size_t)m->Hkv * max_seq * (size_t)m->d;
    c->m = m; c->max_seq = max_seq; c->len = 0;
    c->k = (float *)malloc(cnt * 4); c->v = (float *)malloc(cnt * 4);
    if (!c->k || !c->v) { free(c->k); free(c->v); free(c); FAIL("out of memory"); }
    *out = c; return 0;
}
void orc_cache_reset(orc_cache *c) { c->len = 0; }
size_t orc_cache_len(const orc_cache *c) { return c->len; }
void orc_cache_destroy(orc_cache *c) { if (c) { free(c->k); free(c->v); free(c); } }

/* ---------------------------------------------------------------- kernels */
static inline float dot_f32(const float *w, const float *x, int64_t K) {
    float acc[16] = {0};
    int64_t k = 0;
    for (; k + 16 <= K; k += 16)
        for (int j = 0; j < 16; j++) acc[j] += w[k + j] * x[k + j];
    float s = 0.f;
    for (; k < K; k++) s += w[k] * x[k];
    for (int j = 0; j < 16; j++) s += acc[j];
    return s;
}
static inline float dot_bf16(const uint16_t *w, const float *x, int64_t K) {
    float acc[16] = {0};
    int64_t k = 0;
    for (; k + 16 <= K; k += 16)
        for (int j = 0; j < 16; j++) acc[j] += bf16_to_f32(w[k + j]) * x[k + j];
    float s = 0.f;
    for (; k < K; k++) s += bf16_to_f32(w[k]) * x[k];
    for (int j = 0; j < 16; j++) s += acc[j];
    return s;
}

/* Linear::forward: y[t][n] = sum_k x[t][k] * W[n][k] (+ b[n])   (App. A.1) */
static void linear_fwd(const orc_model *m, const linear_t *lin, const float *x, int64_t T, float *y) {
    const int64_t N = lin->N, K = lin->K;
    int nt = orc_model_threads(m); (void)nt;
    /* tiny matrices: thread start-up would dominate */
#pragma omp parallel for schedule(static) num_threads(nt) if ((double)N * (double)K * (double)T > 2e6)
    for (int64_t n = 0; n < N; n++) {
        float b = lin->bias ? lin->bias[n] : 0.f;
        if (lin->wb) { const uint16_t *w = lin->wb + (size_t)n * K;
            for (int64_t t = 0; t < T; t++) y[(size_t)t * N + n] = dot_bf16(w, x + (size_t)t * K, K) + b;
        } else { const float *w = lin->wf + (size_t)n * K;
            for (int64_t t = 0; t < T; t++) y[(size_t)t * N + n] = dot_f32(w, x + (size_t)t * K, K) + b;
        }
    }
}

/* candle_nn::ops::rms_norm (App. A.2): m = sqrt(sum(x^2)/h + eps); y = x / m * w */
static void rmsnorm_fwd(const orc_model *m, const float *x, const float *w, int64_t T, float *y) {
    const int64_t h = m->h;
    for (int64_t t = 0; t < T; t++) {
        const float *xr = x + (size_t)t * h; float *yr = y + (size_t)t * h;
        float ss = 0.f;
        for (int64_t i = 0; i < h; i++) ss += xr[i] * xr[i];
        float r = sqrtf(ss / (float)h + m->eps);
        if (m->round_bf16 == 2) {               /* candle bf16 (App. A.2): m and every op's result rounded to bf16 */
            const float rb = round_bf16f(r);
            for (int64_t i = 0; i < h; i++) yr[i] = round_bf16f(round_bf16f(xr[i] / rb) * w[i]);
            continue;
        }
        for (int64_t i = 0; i < h; i++) {
            /* bf16 emulation: the MI355X path stores x*w in bf16 and applies 1/rms after the
             * projection's dot product, so the rounding point is x*w */
            yr[i] = m->round_bf16 ? round_bf16f(xr[i] * w[i]) / r : xr[i] / r * w[i];
        }
    }
}

/* candle_nn::rotary_emb::rope, rotate-half form (App. A.4), in place on [T][nh][d] */
static void rope_fwd(const orc_model *m, float *x, int64_t T, int64_t nh, size_t pos) {
    const int64_t d = m->d, half = d / 2;
    for (int64_t t = 0; t < T; t++) {
        float p = (float)(pos + (size_t)t);
        for (int64_t hh = 0; hh < nh; hh++) {
            float *v = x + ((size_t)t * nh + hh) * d;
            for (int64_t j = 0; j < half; j++) {
                float ang = p * m->inv_freq[j];
                float c = cosf(ang), s = sinf(ang);
                if (m->round_bf16 == 2) {
                    /* candle bf16 tables (App. A.4): Llama builds the angles in f32 and casts cos / sin to bf16;
                     * Mistral / Qwen2 cast inv_freq AND the position vector to bf16 before the outer product (positions
                     * above 256 are already rounded) and take cos / sin of the bf16 product */
                    if (m->cfg.family == ORC_LLAMA) { c = round_bf16f(c); s = round_bf16f(s); }
                    else {
                        const float fb = round_bf16f(round_bf16f(p) * round_bf16f(m->inv_freq[j]));
                        c = round_bf16f(cosf(fb)); s = round_bf16f(sinf(fb));
                    }
                }
                float a = v[j], b = v[j + half];
                v[j] = a * c - b * s;
                v[j + half] = a * s + b * c;
            }
        }
    }
}

static void maybe_round(const orc_model *m, float *x, size_t n) {
    if (m->round_bf16) for (size_t i = 0; i < n; i++) x[i] = round_bf16f(x[i]);
}
static void round_all(float *x, size_t n) { for (size_t i = 0; i < n; i++) x[i] = round_bf16f(x[i]); }

/* Causal SDPA over the cache (App. A.3, A.5, A.6).  q [T][H][d]; keys 0..len+T.
 * Mask (only when T > 1): key index kj < len (cached prefix) is visible; for the
 * in-call part j = kj-len, i = query row: masked if j > i, or (Mistral/Qwen2) if
 * j + sliding_window < i.  T == 1: no mask, no window. */
static void attention_fwd(const orc_model *m, const orc_cache *c, int64_t layer, const float *q,
                          int64_t T, size_t len, float *out) {
    const int64_t H = m->H, Hkv = m->Hkv, d = m->d, G = H / Hkv;
    const size_t S = len + (size_t)T;
    const int cmode = m->round_bf16 == 2 && m->cfg.family != ORC_LLAMA;
    int nt = orc_model_threads(m); (void)nt;
#pragma omp parallel for collapse(2) schedule(static) num_threads(nt) if ((double)H * (double)T * (double)S * (double)d > 2e6)
    for (int64_t hq = 0; hq < H; hq++) {
        for (int64_t t = 0; t < T; t++) {
            const int64_t hk = hq / G;                                 /* repeat_kv: consecutive */
            const float *kb = c->k + (((size_t)layer * Hkv + hk) * c->max_seq) * d;
            const float *vb = c->v + (((size_t)layer * Hkv + hk) * c->max_seq) * d;
            const float *qv = q + ((size_t)t * H + hq) * d;
            float *sc = (float *)malloc(S * sizeof(float));
            float mx = -INFINITY;
            for (size_t kj = 0; kj < S; kj++) {
                int masked = 0;
                if (T > 1 && kj >= len) {
                    int64_t j = (int64_t)(kj - len);
                    if (j > t) masked = 1;
                    else if (m->window >= 0 && j + m->window < t) masked = 1;
                }
                float s = -INFINITY;
                if (!masked) {
                    s = dot_f32(kb + kj * d, qv, d);
                    /* candle bf16, Mistral / Qwen2 (App. A.3): the q.k^T matmul and the scaling each round to bf16; Llama
                     * upcasts q, k, v to f32 and divides by sqrt(d) there */
                    if (cmode) s = round_bf16f(round_bf16f(s) * round_bf16f(m->scale)); else s *= m->scale;
                }
                sc[kj] = s; if (s > mx) mx = s;
            }
            float sum = 0.f;
            for (size_t kj = 0; kj < S; kj++) { float e = expf(sc[kj] - mx); sc[kj] = e; sum += e; }
            float *o = out + ((size_t)t * H + hq) * d;
            for (int64_t j = 0; j < d; j++) o[j] = 0.f;
            for (size_t kj = 0; kj < S; kj++) {
                float pw = sc[kj] / sum;
                if (cmode) pw = round_bf16f(pw);                    /* softmax_last_dim returns bf16 probabilities */
                if (pw == 0.f) continue;
                const float *vr = vb + kj * d;
                for (int64_t j = 0; j < d; j++) o[j] += pw * vr[j];
            }
            free(sc);
        }
    }
}

int orc_forward(orc_model *m, orc_cache *c, const uint32_t *ids, size_t T_, size_t pos, float *logits_out) {
    if (!m || !c || !ids || !logits_out) FAIL("null argument");
    if (T_ == 0) FAIL("empty input");
    if (c->m != m) FAIL("cache belongs to another model");
    if (c->len + T_ > c->max_seq) FAIL("sequence overflow: %zu + %zu > %zu", c->len, T_, c->max_seq);
    const int64_t T = (int64_t)T_, h = m->h, H = m->H, Hkv = m->Hkv, d = m->d, I = m->inter;
    for (int64_t t = 0; t < T; t++) if ((int64_t)ids[t] >= m->V) FAIL("token id %u out of range", ids[t]);

    float *x   = (float *)malloc((size_t)T * h * 4);
    float *xn  = (float *)malloc((size_t)T * h * 4);
    float *q   = (float *)malloc((size_t)T * H * d * 4);
    float *k   = (float *)malloc((size_t)T * Hkv * d * 4);
    float *v   = (float *)malloc((size_t)T * Hkv * d * 4);
    float *ao  = (float *)malloc((size_t)T * H * d * 4);
    float *tmp = (float *)malloc((size_t)T * h * 4);
    float *g   = (float *)malloc((size_t)T * I * 4);
    float *u   = (float *)malloc((size_t)T * I * 4);
    if (!x || !xn || !q || !k || !v || !ao || !tmp || !g || !u) {
        free(x); free(xn); free(q); free(k); free(v); free(ao); free(tmp); free(g); free(u); FAIL("out of memory");
    }
    /* Embedding::forward */
    for (int64_t t = 0; t < T; t++)
        for (int64_t i = 0; i < h; i++)
            x[(size_t)t * h + i] = m->embed_b ? bf16_to_f32(m->embed_b[(size_t)ids[t] * h + i]) : m->embed_f[(size_t)ids[t] * h + i];

    const size_t len = c->len;
    /* round_bf16 == 2: candle's bf16 execution, every op's output a bf16 tensor ([UPSTREAM-RECALLED]:
     * the crate is not in /root/reference).  Used only to MEASURE how far the product's bf16 logits are from what the
     * reference's hard-wired BF16 run (main.rs:120) would produce; the parity bar itself is the fp32 mode. */
    const int candle = m->round_bf16 == 2;
    for (int64_t l = 0; l < m->L; l++) {
        const layer_t *ly = &m->layers[l];
        rmsnorm_fwd(m, x, ly->ln1, T, xn);
        linear_fwd(m, &ly->q, xn, T, q);
        linear_fwd(m, &ly->k, xn, T, k);
        linear_fwd(m, &ly->v, xn, T, v);
        if (candle) { round_all(q, (size_t)T * H * d); round_all(k, (size_t)T * Hkv * d); round_all(v, (size_t)T * Hkv * d); }
        rope_fwd(m, q, T, H, pos);
        rope_fwd(m, k, T, Hkv, pos);
        maybe_round(m, q, (size_t)T * H * d); maybe_round(m, k, (size_t)T * Hkv * d); maybe_round(m, v, (size_t)T * Hkv * d);
        /* Tensor::cat(prev, new, dim=2): append at cache len (App. A.8) */
        for (int64_t t = 0; t < T; t++)
            for (int64_t hk = 0; hk < Hkv; hk++) {
                size_t dst = ((((size_t)l * Hkv + hk) * c->max_seq) + len + (size_t)t) * d;
                memcpy(c->k + dst, k + ((size_t)t * Hkv + hk) * d, (size_t)d * 4);
                memcpy(c->v + dst, v + ((size_t)t * Hkv + hk) * d, (size_t)d * 4);
            }
        attention_fwd(m, c, l, q, T, len, ao);
        maybe_round(m, ao, (size_t)T * H * d);
        linear_fwd(m, &ly->o, ao, T, tmp);
        if (m->allreduce) m->allreduce(tmp, (size_t)T * h, m->allreduce_ctx);
        if (candle) round_all(tmp, (size_t)T * h);
        for (size_t i = 0; i < (size_t)T * h; i++) x[i] += tmp[i];
        if (candle) round_all(x, (size_t)T * h);                /* the residual stream itself is a bf16 tensor in candle */
        rmsnorm_fwd(m, x, ly->ln2, T, xn);
        linear_fwd(m, &ly->gate, xn, T, g);
        linear_fwd(m, &ly->up, xn, T, u);
        if (candle) { round_all(g, (size_t)T * I); round_all(u, (size_t)T * I); }
        for (size_t i = 0; i < (size_t)T * I; i++) {       /* silu(g) * u; silu = x / (1 + exp(-x)) */
            float sg = g[i] / (1.0f + expf(-g[i]));
            if (candle) sg = round_bf16f(sg);                   /* Activation::Silu, then the multiply: one rounding each */
            float a = sg * u[i];
            g[i] = m->round_bf16 ? round_bf16f(a) : a;
        }
        linear_fwd(m, &ly->down, g, T, tmp);
        if (m->allreduce) m->allreduce(tmp, (size_t)T * h, m->allreduce_ctx);
        if (candle) round_all(tmp, (size_t)T * h);
        for (size_t i = 0; i < (size_t)T * h; i++) x[i] += tmp[i];
        if (candle) round_all(x, (size_t)T * h);
    }
    c->len = len + (size_t)T;
    /* narrow(1, T-1, 1) -> final norm -> lm_head; logits as f32 (App. A.1) */
    rmsnorm_fwd(m, x + (size_t)(T - 1) * h, m->norm, 1, xn);
    linear_fwd(m, &m->lm_head, xn, 1, logits_out);
    if (candle) round_all(logits_out, (size_t)m->V);            /* the lm_head matmul returns bf16 (Llama widens it afterwards) */
    free(x); free(xn); free(q); free(k); free(v); free(ao); free(tmp); free(g); free(u);
    return 0;
}

/* LogitsProcessor::sample with temperature < 1e-7 => ArgMax (App. A.7); Rust's
 * Iterator::max_by returns the LAST maximal element on ties. */
uint32_t orc_argmax(const float *logits, size_t n) {
    size_t best = 0;
    for (size_t i = 1; i < n; i++) if (!(logits[i] < logits[best])) best = i;
    return (uint32_t)best;
}

/* Model<M>::generate, src/models/mod.rs:363-463 (same loop body as
 * generate_tokens_inner, mod.rs:268-340), temperature 0. */
int orc_generate(orc_model *m, orc_cache *c, const uint32_t *prompt, size_t T, size_t max_tokens,
                 int64_t eos, int pos_mode, uint32_t *out_tokens, float *step_logits) {
    const size_t V = (size_t)m->V;
    float *logits = (float *)malloc(V * 4);
    if (!logits) FAIL("out of memory");
    orc_cache_reset(c);                         /* mod.rs:370 fresh cache per request */
    size_t counter = 0;                         /* MistralCache/QwenCache.seqlen_offset */
    const int use_counter = pos_mode == 1 && m->cfg.family != ORC_LLAMA;
    size_t pos = 0;
    if (orc_forward(m, c, prompt, T, use_counter ? counter : pos, logits)) { free(logits); return -1; }
    counter++;                                  /* mistral.rs:234: +1 per call, not +T */
    pos += T;                                   /* mod.rs:408 */
    int n_out = 0;
    for (size_t i = 0; i < max_tokens; i++) {   /* mod.rs:411 */
        uint32_t tok = orc_argmax(logits, V);
        if (eos >= 0 && (int64_t)tok == eos) break;          /* mod.rs:431-436: stop before emitting */
        if (step_logits) memcpy(step_logits + (size_t)n_out * V, logits, V * 4);
        out_tokens[n_out++] = tok;
        /* mod.rs:446-452: the forward also runs after the last kept token (quirk C.5) */
        if (c->len + 1 > c->max_seq) break;
        if (orc_forward(m, c, &tok, 1, use_counter ? counter : pos, logits)) { free(logits); return -1; }
        counter++; pos += 1;
    }
    free(logits);
    return n_out;
}
