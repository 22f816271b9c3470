// model.hip -- model build (weight sharding + re-layout into HBM), KV cache, and the
// per-layer orchestration of the decoder forward pass on MI355X.
//
// Reference semantics reproduced here (all citations into /root/reference/src/models):
//   config defaults + validation   llama.rs:31-50, mistral.rs:93-154, qwen.rs:30-56, config.rs:31-54
//   weights bound by HF name       llama.rs:112-120, mistral.rs:190-192, qwen.rs:108-109
//   forward(input, pos, cache)     llama.rs:147-149, mistral.rs:206-236, qwen.rs:123-151
// The arithmetic follows candle 0.8.x.
//
// HBM layout (per shard; compute dtype = bf16 or fp32):
//   wqkv [(Hs+2Hkvs)d, h]  fused q|k|v rows       wo [h, Hs*d]
//   wgu  [2*Ip, h] gate/up rows interleaved 16x16  wd [h, Ip]      (Ip = Is rounded up to 16)
//   lm_head [Vs, h], embed [V, h], norms fp32, RoPE cos/sin fp32 [max_pos][d/2]
//   KV cache [L][Hkvs][max_seq][d] x2, written in place (no Tensor::cat copy)
#include "model.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <ctype.h>

#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>

namespace fl {

// ------------------------------------------------------------------------------- errors
static thread_local char g_err[1024];
void set_error(const char *fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap);
}
const char *last_error() { return g_err; }


// the library's only reads of the environment: the switch table below (integers) and a handful of diagnostic paths / the fault injector (strings)
const char *env_str(const char *name) { const char *s = getenv(name); return s && *s ? s : nullptr; }
int env_int(const char *name, int dflt) { const char *s = env_str(name); return s ? atoi(s) : dflt; }

// ------------------------------------------------------------------------------- switches (common.h: TuneKey)
struct TuneEntry { const char *name; int dflt; bool exp_only = false; };   // exp_only: acts in the EXPERIMENTAL build only (Makefile); fl_tune refuses it elsewhere
static const TuneEntry g_tune_table[TK_COUNT] = {
    {"gemm_h4", 1},
    {"gemm_w14", 1},
    {"gemm_rope_4w", 1},
    {"gemm_f32_mfma", 1},
    {"w14_nt", -1},
    {"h4_nt", -1},
    {"h4_split", 0},
    {"h4_pf", 6},
    {"h4_wait_us", 30},
    {"op_maxsplit", 0},
    {"op_linear_dma", 0},
    {"op_hot", 0},
    {"ar_inbox_floats", 131072},
    {"ar_timeout_ms", 20000},
    {"verbose", 0},
    {"tp_fused_ar", 1},
    {"attn_nw", 4},
    {"attn_prefetch", 0, true},
    {"attn_prefetch_lines", 8, true},
    {"attn_prefetch_pct", 100, true},
    {"attn_prefetch_delay", 0, true},
    {"attn_batch_wgs", 256},
    {"attn_pf32_min_t", 0},
    {"attn_pf32_ks2", -1},
    {"attn_pf32_paired", -1},
    {"attn_pf_waves", 0},
    {"attn_pf_stages", 2},
    {"attn_pf_ksplit", 2},
    {"ao_delay", 6, true},
    {"ao_waves", 0, true},
    {"engine_delay", 12, true},
    {"engine_pf", 1, true},
    {"engine_timeout_ms", 2000, true},
    {"sk_minsteps", 8},
    {"gemm_4w", 1},
    {"gemm_groupm", 0},
    {"8p_mink", 8},
    {"gemm_8p", 1},
    {"gemm_256", 1},
    {"gemm_256_split", 1},
    {"gemm_streamk", 1},
    {"gemm_peel", 1},
    {"gemm_resid", 1},
    {"gemm_skinny_maxt", 128},
    {"skinny_stages", 4},
    {"skinny_nt", 1},
    {"skinny_wm", 1},
    {"skinny_loaders", -1, true},
    {"gemm_skinny_maxt2", 256},
    {"gemv_small", 1},
    {"gemv_r", 2},
    {"gemv_u", 0},
    {"batch_u", 0},
    {"batch_mode", 2},
    {"batch_mfma_min", 3},
    {"dma_kt", 128},
    {"force_generic_gemm", 0},
    {"gemm_skinny", 1},
    {"rope_vec", 1},
    {"weight_arena", 1},
    {"ksplit_mid", 0},
    {"prefill_chunk", 8192},
    {"graph", -1},
    {"fused", 1},
    {"allow_any_arch", 0},
    {"engine", 0, true},
    {"fuse_oproj", 0, true},
    {"oneshot", 1},
    {"debug_rccl_self", 0},
    {"attn_mfma", 1},
    {"attn_nsplit", -1},
    {"attn_rep", 1},
    {"sample_walk", 0},
    {"argmax_fused", 1},
    {"tp_overlap", 1},
    {"tp_overlap_min_t", 512},
    {"qkv_split", 8},
    {"tp_graph", 1},
    {"batch_dma_min", 3},
    {"h4_oproj_1k", 1},
    {"h4_tail", 2},
    {"rs_lazy", 1},
    {"batch_unfused_min", -1},
    {"debug_rs_parts", 0},
    {"debug_tp_loopback", 0, true},      // (results are meaningless by design: a timing tool of the EXPERIMENTAL build)
    {"debug_poison", 0},
    {"gemm_skf", 1},
    {"skf_split", 0},
    {"prefill_dma", 1},
    {"oneshot_wide", 1},
    {"f32_rows_max", 64},
    {"gateup_rowsplit", 1},
};
static_assert(sizeof(g_tune_table) / sizeof(g_tune_table[0]) == TK_COUNT, "one row per TuneKey, in the enum's order");
static std::atomic<int> g_tune[TK_COUNT];
static std::once_flag g_tune_once;
static void tune_read_env() {
    for (int k = 0; k < TK_COUNT; k++) {
        char env[64] = "FL_";
        size_t n = 3;
        for (const char *c = g_tune_table[k].name; *c && n + 1 < sizeof env; c++) env[n++] = (char)toupper((unsigned char)*c);
        env[n] = 0;
        int v = env_int(env, g_tune_table[k].dflt);
#ifndef FL_EXPERIMENTAL
        if (g_tune_table[k].exp_only) v = g_tune_table[k].dflt;       // the environment cannot reach a kernel that is not compiled in
        if (k == TK_H4_PF) v &= 0xFFFF;
#endif
        g_tune[k].store(v, std::memory_order_relaxed);
    }
}
int tune(TuneKey k) {
    std::call_once(g_tune_once, tune_read_env);
    return g_tune[k].load(std::memory_order_relaxed);
}
void tune_poison_restart();
int tune_set(const char *name, int value) {
    std::call_once(g_tune_once, tune_read_env);
    for (int k = 0; k < TK_COUNT; k++)
        if (!strcmp(name, g_tune_table[k].name)) {
#ifndef FL_EXPERIMENTAL
            if (g_tune_table[k].exp_only) return FL_ERR_UNSUPPORTED;
            if (k == TK_H4_PF) value &= 0xFFFF;      // (bit 16 is a wrong-results timing probe of the experimental build)
#endif
            if (k == TK_DEBUG_POISON) tune_poison_restart();
            g_tune[k].store(value, std::memory_order_relaxed);
            return FL_OK;
        }
    return FL_ERR_BAD_ARGUMENT;
}
void tune_reload_env() {
    std::call_once(g_tune_once, tune_read_env);
    tune_read_env();
}

int raise_dynamic_lds(const void *fn, size_t lds) {
    if (lds < 64 * 1024) return FL_OK;
    static std::mutex mu;
    static std::unordered_map<uint64_t, size_t> raised;           // (function, device) -> bytes granted
    int dev = 0;
    FL_HIP(hipGetDevice(&dev));
    const uint64_t key = (uint64_t)(uintptr_t)fn * 64 + (uint64_t)(dev & 63);
    std::lock_guard<std::mutex> lock(mu);
    auto it = raised.find(key);
    if (it != raised.end() && it->second >= lds) return FL_OK;
    FL_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    raised[key] = lds;
    return FL_OK;
}

// Fault injection for the tests of the ABI's exception barrier: FL_DEBUG_THROW="<site>=<bad_alloc|runtime|int>"
// makes the named site throw, as a failed `new` / std::vector growth would.
void debug_inject(const char *site) {
    const char *s = env_str("FL_DEBUG_THROW");
    if (!s || !*s) return;
    const size_t n = strlen(site);
    if (strncmp(s, site, n) || s[n] != '=') return;
    if (!strcmp(s + n + 1, "bad_alloc")) throw std::bad_alloc();
    if (!strcmp(s + n + 1, "runtime")) throw std::runtime_error("injected failure");
    throw 42;
}
struct PeerComm;

// ------------------------------------------------------------------------------- config
int resolve_config(const fl_config *cfg, Dims *o) {
    if (!cfg) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null config");
    Dims D;
    D.family = cfg->family;
    if (D.family < FL_FAMILY_LLAMA || D.family > FL_FAMILY_QWEN2) FL_FAIL(FL_ERR_BAD_CONFIG, "unknown model family %d", D.family);
    D.qkv_bias = cfg->qkv_bias != 0;
    D.h = cfg->hidden_size; D.inter = cfg->intermediate_size; D.V = cfg->vocab_size;
    D.L = cfg->num_hidden_layers; D.H = cfg->num_attention_heads;
    if (D.h <= 0 || D.inter <= 0 || D.V <= 0 || D.L <= 0 || D.H <= 0) FL_FAIL(FL_ERR_BAD_CONFIG, "non-positive model dimension");
    D.Hkv = cfg->num_key_value_heads > 0 ? cfg->num_key_value_heads : D.H;          // llama.rs:39
    D.dm = D.h / D.H;
    if (D.dm * D.H != D.h) FL_FAIL(FL_ERR_BAD_CONFIG, "hidden_size must be divisible by num_attention_heads");   // config.rs:34
    if (D.dm % 2) FL_FAIL(FL_ERR_BAD_CONFIG, "head_dim must be even for RoPE embeddings");                       // config.rs:39
    // The kernels are built for head_dim 64 and 128 (MFMA tiles, 16-byte rows).  Any other even head_dim up to 128 -- the reference
    // takes every even value (config.rs:31-43; e.g. 80, 96, 100) -- runs as the next of the two: every head's q / k / v rows are
    // laid out as [first half | zeros | second half | zeros] (so rotate-half pairs stay dm/2... d/2 apart) and o_proj gets zero
    // columns to match; the padded lanes carry exact zeros through RoPE, scores and values.  Above 128: fl_model_create refuses.
    D.d = D.dm <= 64 ? 64 : 128;
    if (D.H % D.Hkv) FL_FAIL(FL_ERR_BAD_CONFIG, "num_attention_heads must be divisible by num_key_value_heads"); // config.rs:48
    if (cfg->rms_norm_eps < 0) FL_FAIL(FL_ERR_BAD_CONFIG, "negative rms_norm_eps");
    D.eps = (float)cfg->rms_norm_eps;
    D.theta = cfg->rope_theta > 0 ? cfg->rope_theta : 10000.0;                                                   // llama.rs:41
    const int64_t dflt_pos = D.family == FL_FAMILY_LLAMA ? 4096 : 32768;                                        // llama.rs:47, mistral.rs:138
    D.max_pos = cfg->max_position_embeddings > 0 ? cfg->max_position_embeddings : dflt_pos;
    if (D.family == FL_FAMILY_LLAMA) D.window = -1;
    else D.window = cfg->sliding_window > 0 ? cfg->sliding_window : (cfg->sliding_window < 0 ? -1 : 4096);      // mistral.rs:139
    D.scale = (float)(1.0 / sqrt((double)D.dm));                                                                 // (the model's head_dim, not the padded one)
    *o = D;
    return FL_OK;
}

static bool ends_with(const std::string &s, const char *suf) {
    size_t n = strlen(suf); return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// Megatron-style partition: q/k/v/gate/up/lm_head column-parallel (rows of the
// [out,in] matrix), o_proj/down_proj row-parallel (columns); norms and the embedding whole.
int tp_slice(const Dims &D, const char *name_c, int rank, int tp, int64_t out[4]) {
    if (tp < 1 || rank < 0 || rank >= tp) FL_FAIL(FL_ERR_BAD_ARGUMENT, "bad tp rank %d of %d", rank, tp);
    if (D.H % tp || D.Hkv % tp) FL_FAIL(FL_ERR_UNSUPPORTED, "tp=%d must divide heads (%lld) and kv heads (%lld)", tp, (long long)D.H, (long long)D.Hkv);
    if (D.inter % tp) FL_FAIL(FL_ERR_UNSUPPORTED, "tp=%d must divide intermediate_size %lld", tp, (long long)D.inter);
    const std::string name(name_c);
    const int64_t qd = D.H * D.dm, kvd = D.Hkv * D.dm;             // (source tensor coordinates: the model's head_dim)
    int64_t R = 0, C = 0, r0 = 0, r1 = 0, c0 = 0, c1 = 0;
    auto rows = [&](int64_t n, int64_t k) { R = n; C = k; r0 = n / tp * rank; r1 = n / tp * (rank + 1); c0 = 0; c1 = k; };
    auto cols = [&](int64_t n, int64_t k) { R = n; C = k; r0 = 0; r1 = n; c0 = k / tp * rank; c1 = k / tp * (rank + 1); };
    auto whole = [&](int64_t n, int64_t k) { R = n; C = k; r0 = 0; r1 = n; c0 = 0; c1 = k; };
    if (ends_with(name, "q_proj.weight")) rows(qd, D.h);
    else if (ends_with(name, "k_proj.weight") || ends_with(name, "v_proj.weight")) rows(kvd, D.h);
    else if (ends_with(name, "q_proj.bias")) rows(qd, 1);
    else if (ends_with(name, "k_proj.bias") || ends_with(name, "v_proj.bias")) rows(kvd, 1);
    else if (ends_with(name, "o_proj.weight")) cols(D.h, qd);
    else if (ends_with(name, "gate_proj.weight") || ends_with(name, "up_proj.weight")) rows(D.inter, D.h);
    else if (ends_with(name, "down_proj.weight")) cols(D.h, D.inter);
    else if (name == "lm_head.weight") { if (D.V % tp == 0) rows(D.V, D.h); else whole(D.V, D.h); }
    else if (name == "model.embed_tokens.weight") whole(D.V, D.h);
    else if (ends_with(name, "layernorm.weight") || name == "model.norm.weight") whole(D.h, 1);
    else FL_FAIL(FL_ERR_MISSING_TENSOR, "unknown tensor name %s", name_c);
    (void)R; (void)C;
    out[0] = r0; out[1] = r1; out[2] = c0; out[3] = c1;
    return FL_OK;
}

// ------------------------------------------------------------------------------- allocation
static std::atomic<int> g_poison_count{0};
void tune_poison_restart() { g_poison_count.store(0); }
static int dev_alloc(std::vector<void *> &owner, void **p, size_t bytes, int64_t *acct) {
    if (bytes == 0) bytes = 16;
    const hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); FL_FAIL(FL_ERR_OOM, "out of HBM: hipMalloc of %zu bytes failed", bytes); }
    FL_HIP(e);
    owner.push_back(*p);
    if (acct) *acct += (int64_t)bytes;
    if (const int fill = tune(TK_DEBUG_POISON)) {
        // bits 0-7: the byte; bits 8+: 0 = every allocation, n = only the n-th since the switch was last set (tools/poison_probe.py scans)
        const int nth = g_poison_count.fetch_add(1) + 1, want = fill >> 8;
        if (want == 0 || want == nth) { FL_HIP(hipMemset(*p, fill & 0xFF, bytes)); FL_HIP(hipDeviceSynchronize()); }
    }
    return FL_OK;
}

Model::~Model() {
    std::vector<hipStream_t> closed;            // EMULATED shards share one stream
    for (auto &s : shards) {
        (void)hipSetDevice(s.device);
        if (s.stream) (void)hipStreamSynchronize(s.stream);
        if (s.comm) ncclCommDestroy(s.comm);
        if (s.comm_stream) { (void)hipStreamSynchronize(s.comm_stream); (void)hipStreamDestroy(s.comm_stream); }
        for (auto &e : s.ev) if (e) (void)hipEventDestroy(e);
        for (void *mp : s.pc.mapped) if (mp) (void)hipIpcCloseMemHandle(mp);
        if (s.pc.local) { comm_forget(s.pc.local); comm_inbox_release(s.device, s.pc.bytes, s.pc.local); }
        if (s.pc.epoch) (void)hipFree(s.pc.epoch);
        if (s.pc.ll_dev) (void)hipFree(s.pc.ll_dev);
        if (s.pc.err) (void)hipHostFree(s.pc.err);
        for (void *p : s.allocs) (void)hipFree(p);
        for (void *p : s.pre_allocs) (void)hipFree(p);
        if (s.stream && std::find(closed.begin(), closed.end(), s.stream) == closed.end()) {
            closed.push_back(s.stream);
            gemm_8p_release_stream(s.stream);
            gemm_h4_release_stream(s.stream);
            gemm_skf_release_stream(s.stream);
            (void)hipStreamDestroy(s.stream);
        }
    }
    if (emu_ptrs) (void)hipFree(emu_ptrs);
    if (host_logits) (void)hipHostFree(host_logits);
    if (host_tokens) (void)hipHostFree(host_tokens);
    if (host_state) (void)hipHostFree(host_state);
    for (auto &r : prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
}

Cache::~Cache() {
    if (!m) return;
    std::lock_guard<std::mutex> lock(m->mu);       // not while another thread captures on the model's stream
    for (size_t i = 0; i < shards.size(); i++) {
        (void)hipSetDevice(m->shards[i].device);
        (void)hipStreamSynchronize(m->shards[i].stream);
        if (shards[i].graph) (void)hipGraphExecDestroy(shards[i].graph);
        for (void *p : shards[i].allocs) (void)hipFree(p);
    }
}

// ------------------------------------------------------------------------------- weight build
struct Stager {                      // brings a source tensor to a device (whole), reusing one buffer
    int device; void *buf = nullptr; size_t cap = 0;
    explicit Stager(int dev) : device(dev) {}
    ~Stager() { if (buf) { (void)hipSetDevice(device); (void)hipFree(buf); } }
    int get(const fl_tensor &t, size_t bytes, const void **out) {
        if (t.device == device) { *out = t.data; return FL_OK; }
        if (bytes > cap) {
            if (buf) { FL_HIP(hipFree(buf)); buf = nullptr; cap = 0; }
            FL_HIP(hipMalloc(&buf, bytes)); cap = bytes;
        }
        FL_HIP(hipMemcpy(buf, t.data, bytes, hipMemcpyDefault));
        *out = buf;
        return FL_OK;
    }
};

static size_t dtype_size(int dt) { return dt == FL_DTYPE_F32 ? 4 : 2; }

struct Builder {
    Model *m;
    std::unordered_map<std::string, const fl_tensor *> map;
    const fl_tensor *find(const std::string &name) const {
        auto it = map.find(name); return it == map.end() ? nullptr : it->second;
    }
    int want(const std::string &name, int64_t R, int64_t C, const fl_tensor **out) const {
        const fl_tensor *t = find(name);
        if (!t) FL_FAIL(FL_ERR_MISSING_TENSOR, "cannot find tensor %s", name.c_str());
        if (t->dtype < FL_DTYPE_F32 || t->dtype > FL_DTYPE_F16) FL_FAIL(FL_ERR_UNSUPPORTED, "tensor %s: unsupported dtype %d", name.c_str(), t->dtype);
        bool ok = (C == 1 && t->ndim == 1) ? t->shape[0] == R : (t->ndim == 2 && t->shape[0] == R && t->shape[1] == C);
        if (!ok) FL_FAIL(FL_ERR_SHAPE_MISMATCH, "shape mismatch for %s: expected [%lld,%lld]", name.c_str(), (long long)R, (long long)C);
        if (!t->data) FL_FAIL(FL_ERR_BAD_ARGUMENT, "tensor %s has null data", name.c_str());
        *out = t; return FL_OK;
    }
};

// Copy slice [r0,r1) x [c0,c1) of tensor `name` (full shape R x C) into dst (ld = dst_ld) on every
// shard that lives on st.device; dst_of(shard) gives the destination base, row_mode the row map.
// head_pad: 0 none; 1 the ROWS are heads of the model's head_dim dm, placed as padded heads of d rows; 2 the COLUMNS are
template <typename DstFn>
static int put_matrix(Builder &B, Stager &st, const std::string &name, int64_t R, int64_t C, int dst_dtype,
                      int64_t dst_ld, int64_t dst_row0, int row_mode, DstFn dst_of, int head_pad = 0) {
    Model *m = B.m;
    const fl_tensor *t = nullptr;
    FL_TRY(B.want(name, R, C, &t));
    const void *src = nullptr;
    FL_TRY(st.get(*t, (size_t)R * C * dtype_size(t->dtype), &src));
    for (auto &sh : m->shards) {
        if (sh.device != st.device) continue;
        int64_t sl[4];
        FL_TRY(tp_slice(m->D, name.c_str(), sh.rank, m->tp, sl));
        Launcher L; L.stream = sh.stream;
        FL_TRY(launch_convert_slice(L, t->dtype, src, C, sl[0], sl[2], sl[1] - sl[0], sl[3] - sl[2], dst_dtype,
                                    dst_of(sh), dst_ld, dst_row0, row_mode, m->D.dm != m->D.d ? head_pad : 0, m->D.dm, m->D.d));
    }
    FL_HIP(hipDeviceSynchronize());       // the staging buffer is reused by the next tensor
    return FL_OK;
}

static int build_weights(Builder &B) {
    Model *m = B.m;
    const Dims &D = m->D;
    const int wdt = m->dtype;
    const size_t es = m->esize();
    std::vector<int> devices;
    for (auto &sh : m->shards) if (std::find(devices.begin(), devices.end(), sh.device) == devices.end()) devices.push_back(sh.device);

    // allocate: one arena per shard (FL_WEIGHT_ARENA=0: one hipMalloc per tensor).  The whole model is then a single
    // virtual range, which the driver can map with its largest page fragments
    const int use_arena = tune(TK_WEIGHT_ARENA);
    for (auto &sh : m->shards) {
        FL_HIP(hipSetDevice(sh.device));
        const int64_t nq = (sh.Hs + 2 * sh.Hkvs) * D.d;
        char *arena = nullptr; size_t arena_off = 0, arena_cap = 0;
        auto walloc = [&](void **p, size_t bytes) -> int {
            if (!arena) return dev_alloc(sh.allocs, p, bytes, &m->hbm_bytes);
            const size_t a = (bytes + 4095) & ~(size_t)4095;
            if (arena_off + a > arena_cap) FL_FAIL(FL_ERR_OOM, "weight arena too small");
            *p = arena + arena_off; arena_off += a;
            return FL_OK;
        };
        if (use_arena) {
            auto r4k = [](size_t b) { return (b + 4095) & ~(size_t)4095; };
            size_t need = r4k((size_t)D.V * D.h * es) + r4k((size_t)D.h * 4) + r4k((size_t)sh.Vs * D.h * es);
            need += (size_t)D.L * (r4k((size_t)nq * D.h * es) + r4k((size_t)nq * 4) + r4k((size_t)D.h * sh.Hs * D.d * es) +
                                   r4k((size_t)2 * sh.Ip * D.h * es) + r4k((size_t)D.h * sh.Ip * es) + 2 * r4k((size_t)D.h * 4));
            FL_TRY(dev_alloc(sh.allocs, (void **)&arena, need, &m->hbm_bytes));
            arena_cap = need;
        }
        FL_TRY(walloc(&sh.embed, (size_t)D.V * D.h * es));
        FL_TRY(walloc((void **)&sh.norm, (size_t)D.h * 4));
        FL_TRY(walloc(&sh.lm_head, (size_t)sh.Vs * D.h * es));
        sh.layers.resize(D.L);
        for (auto &ly : sh.layers) {
            FL_TRY(walloc(&ly.wqkv, (size_t)nq * D.h * es));
            if (D.qkv_bias) FL_TRY(walloc((void **)&ly.bqkv, (size_t)nq * 4));
            FL_TRY(walloc(&ly.wo, (size_t)D.h * sh.Hs * D.d * es));
            FL_TRY(walloc(&ly.wgu, (size_t)2 * sh.Ip * D.h * es));
            FL_TRY(walloc(&ly.wd, (size_t)D.h * sh.Ip * es));
            FL_TRY(walloc((void **)&ly.ln1, (size_t)D.h * 4));
            FL_TRY(walloc((void **)&ly.ln2, (size_t)D.h * 4));
            if (D.dm != D.d) {               // padded head_dim: the rows / columns between the halves of every head stay zero
                FL_HIP(hipMemsetAsync(ly.wqkv, 0, (size_t)nq * D.h * es, sh.stream));
                if (D.qkv_bias) FL_HIP(hipMemsetAsync(ly.bqkv, 0, (size_t)nq * 4, sh.stream));
                FL_HIP(hipMemsetAsync(ly.wo, 0, (size_t)D.h * sh.Hs * D.d * es, sh.stream));
            }
            if (sh.Ip != sh.Is) {            // zero padding rows/cols so they contribute nothing
                FL_HIP(hipMemsetAsync(ly.wgu, 0, (size_t)2 * sh.Ip * D.h * es, sh.stream));
                FL_HIP(hipMemsetAsync(ly.wd, 0, (size_t)D.h * sh.Ip * es, sh.stream));
            }
        }
        FL_HIP(hipStreamSynchronize(sh.stream));
    }

    const bool has_lm_head = B.find("lm_head.weight") != nullptr;
    if (!has_lm_head && D.family != FL_FAMILY_QWEN2) FL_FAIL(FL_ERR_MISSING_TENSOR, "cannot find tensor lm_head.weight");

    for (int dev : devices) {
        FL_HIP(hipSetDevice(dev));
        Stager st(dev);
        FL_TRY(put_matrix(B, st, "model.embed_tokens.weight", D.V, D.h, wdt, D.h, 0, 0, [](Shard &s) { return s.embed; }));
        FL_TRY(put_matrix(B, st, "model.norm.weight", D.h, 1, FL_DTYPE_F32, 1, 0, 0, [](Shard &s) { return (void *)s.norm; }));
        if (has_lm_head) {
            FL_TRY(put_matrix(B, st, "lm_head.weight", D.V, D.h, wdt, D.h, 0, 0, [](Shard &s) { return s.lm_head; }));
        } else {
            // candle qwen2 falls back to the embedding matrix when lm_head.weight is absent (App. A.1)
            const fl_tensor *t = nullptr; const void *src = nullptr;
            FL_TRY(B.want("model.embed_tokens.weight", D.V, D.h, &t));
            FL_TRY(st.get(*t, (size_t)D.V * D.h * dtype_size(t->dtype), &src));
            for (auto &sh : m->shards) {
                if (sh.device != dev) continue;
                Launcher L; L.stream = sh.stream;
                FL_TRY(launch_convert_slice(L, t->dtype, src, D.h, sh.v0, 0, sh.Vs, D.h, wdt, sh.lm_head, D.h, 0, 0));
            }
            FL_HIP(hipDeviceSynchronize());
        }
        for (int64_t l = 0; l < D.L; l++) {
            const std::string p = "model.layers." + std::to_string(l) + ".";
            auto LY = [l](Shard &s) -> LayerW & { return s.layers[l]; };
            const int64_t qd = D.H * D.dm, kvd = D.Hkv * D.dm;       // (source tensors: the model's head_dim)
            // fused q|k|v: destination row offsets inside the shard's fused matrix
            struct { const char *nm; int64_t R; int which; } qkv[3] = {{"self_attn.q_proj", qd, 0}, {"self_attn.k_proj", kvd, 1}, {"self_attn.v_proj", kvd, 2}};
            for (auto &e : qkv) {
                // all local shards have equal Hs / Hkvs, so the row offset is shard-independent
                const Shard &s0 = m->shards[0];
                const int64_t off = e.which == 0 ? 0 : (e.which == 1 ? s0.Hs * D.d : (s0.Hs + s0.Hkvs) * D.d);
                FL_TRY(put_matrix(B, st, p + e.nm + ".weight", e.R, D.h, wdt, D.h, off, 0, [&](Shard &s) { return LY(s).wqkv; }, 1));
                if (D.qkv_bias)
                    FL_TRY(put_matrix(B, st, p + e.nm + ".bias", e.R, 1, FL_DTYPE_F32, 1, off, 0, [&](Shard &s) { return (void *)LY(s).bqkv; }, 1));
            }
            FL_TRY(put_matrix(B, st, p + "self_attn.o_proj.weight", D.h, qd, wdt, m->shards[0].Hs * D.d, 0, 0, [&](Shard &s) { return LY(s).wo; }, 2));
            FL_TRY(put_matrix(B, st, p + "mlp.gate_proj.weight", D.inter, D.h, wdt, D.h, 0, 1, [&](Shard &s) { return LY(s).wgu; }));
            FL_TRY(put_matrix(B, st, p + "mlp.up_proj.weight", D.inter, D.h, wdt, D.h, 0, 2, [&](Shard &s) { return LY(s).wgu; }));
            FL_TRY(put_matrix(B, st, p + "mlp.down_proj.weight", D.h, D.inter, wdt, m->shards[0].Ip, 0, 0, [&](Shard &s) { return LY(s).wd; }));
            FL_TRY(put_matrix(B, st, p + "input_layernorm.weight", D.h, 1, FL_DTYPE_F32, 1, 0, 0, [&](Shard &s) { return (void *)LY(s).ln1; }));
            FL_TRY(put_matrix(B, st, p + "post_attention_layernorm.weight", D.h, 1, FL_DTYPE_F32, 1, 0, 0, [&](Shard &s) { return (void *)LY(s).ln2; }));
        }
    }
    return FL_OK;
}

// RoPE tables (App. A.4): inv_freq[j] = 1 / theta^(2j/d) in fp32; angle = p * inv_freq[j] (fp32
// product); cos/sin in fp32.  Built once on the host, one copy per shard.
static int build_rope(Model *m) {
    const Dims &D = m->D;
    const int64_t half = D.d / 2, half_m = D.dm / 2;             // pairs of the padded layout; of them, the model's (the rest rotate zeros: identity)
    std::vector<float> inv(half), c((size_t)D.max_pos * half), s((size_t)D.max_pos * half);
    const float theta = (float)D.theta;
    for (int64_t j = 0; j < half_m; j++) inv[j] = 1.0f / powf(theta, (float)(2 * j) / (float)D.dm);
    for (int64_t p = 0; p < D.max_pos; p++)
        for (int64_t j = 0; j < half; j++) {
            const float ang = j < half_m ? (float)p * inv[j] : 0.0f;
            c[(size_t)p * half + j] = cosf(ang);
            s[(size_t)p * half + j] = sinf(ang);
        }
    for (auto &sh : m->shards) {
        FL_HIP(hipSetDevice(sh.device));
        FL_TRY(dev_alloc(sh.allocs, (void **)&sh.cos_tab, c.size() * 4, &m->hbm_bytes));
        FL_TRY(dev_alloc(sh.allocs, (void **)&sh.sin_tab, s.size() * 4, &m->hbm_bytes));
        FL_HIP(hipMemcpy(sh.cos_tab, c.data(), c.size() * 4, hipMemcpyHostToDevice));
        FL_HIP(hipMemcpy(sh.sin_tab, s.data(), s.size() * 4, hipMemcpyHostToDevice));
    }
    return FL_OK;
}

// K slices (fp32 slabs that the next launch -- rmsnorm_add, rope_kv -- sums) a projection may use at T tokens.  Mid-size
// prompts (T = 256..1024: 1-4 row tiles of 256) get up to eight: on the 256x256 kernel Mistral-7B's T = 512 QKV takes
// 47 -> 34.5 us at 5 slices of 12.8 K steps, down_proj 78 -> 59 us at 8 (tools/gemm_probe.py), for ~5 us more in each summing launch.
constexpr int kMaxKSplit = 4;
constexpr int kMaxKSplitMid = 8;
constexpr int kMidT = 1024;
constexpr int kMaxQkvSplit = 2;   // QKV projection of a long prompt (its grid leaves CUs idle); rope_kv sums the slabs
constexpr int kMaxQkvSplitShort = 4;   // ... of a short prompt / a decode batch (T <= 128: the projection is a weight stream)
static int mid_cap(int dflt) { const int v = tune(TK_KSPLIT_MID); return v > 0 ? std::min(v, kMaxKSplitMid) : dflt; }   // (read per call: A/B tools lower it on a live model; the slabs were sized for the default)
int ksplit_cap(int64_t T) { return T <= 1 ? 1 : (T > 128 && T <= kMidT ? mid_cap(kMaxKSplitMid) : kMaxKSplit); }
static int qkv_split_cap(int64_t T) { return T <= 1 ? 1 : (T <= 128 ? kMaxQkvSplitShort : (T <= kMidT ? mid_cap(kMaxKSplitMid) : kMaxQkvSplit)); }
// the same caps with the switch at its largest value: what the slab buffers are SIZED for (a later, larger FL_KSPLIT_MID must
// never write past a buffer that was allocated while it was lowered)
static int ksplit_cap_max(int64_t T) { return T <= 1 ? 1 : (T > 128 && T <= kMidT ? kMaxKSplitMid : kMaxKSplit); }
static int qkv_split_cap_max(int64_t T) { return T <= 1 ? 1 : (T <= 128 ? kMaxQkvSplitShort : (T <= kMidT ? kMaxKSplitMid : kMaxQkvSplit)); }
// rows of slab storage that serve every prompt of at most T tokens
static int64_t slab_rows(int64_t T, int (*cap)(int64_t)) {
    return std::max<int64_t>({T * cap(T), std::min<int64_t>(T, kMidT) * cap(std::min<int64_t>(T, kMidT)), std::min<int64_t>(T, 128) * cap(std::min<int64_t>(T, 128))});
}

// owner: who frees the buffers (default: the shard, i.e. at model destruction)
static int alloc_scratch(Model *m, Shard &sh, Scratch &sc, int64_t T, std::vector<void *> *owner = nullptr) {
    std::vector<void *> &own = owner ? *owner : sh.allocs;
    int64_t *acct = (owner && owner != &sh.pre_allocs) ? nullptr : &m->hbm_bytes;
    const Dims &D = m->D;
    const size_t es = m->esize();
    const int64_t nq = (sh.Hs + 2 * sh.Hkvs) * D.d;
    sc.cap_T = T;
    FL_TRY(dev_alloc(own, (void **)&sc.x_res, (size_t)T * D.h * 4, acct));
    if (T == 1) FL_TRY(dev_alloc(own, (void **)&sc.x_res2, (size_t)D.h * 4, acct));
    // split-K slabs of any prompt <= T; decode: one partial vector per kv head (fused attention + o_proj launch)
    FL_TRY(dev_alloc(own, (void **)&sc.delta, (size_t)std::max<int64_t>(slab_rows(T, ksplit_cap_max), T == 1 ? sh.Hkvs : 0) * D.h * 4, acct));
    FL_TRY(dev_alloc(own, &sc.xn, (size_t)T * D.h * es, acct));
    FL_TRY(dev_alloc(own, (void **)&sc.inv_rms, (size_t)T * 4, acct));
    if (T > 1) FL_TRY(dev_alloc(own, (void **)&sc.rs_part, (size_t)T * gemm_resid_partials(D.h) * 4, acct));
    FL_TRY(dev_alloc(own, (void **)&sc.qkv, (size_t)slab_rows(T, qkv_split_cap_max) * nq * 4, acct));    // split-K slabs of any prompt <= T
    FL_TRY(dev_alloc(own, &sc.q, (size_t)T * sh.Hs * D.d * es, acct));
    FL_TRY(dev_alloc(own, &sc.ao, (size_t)T * sh.Hs * D.d * es, acct));
    FL_TRY(dev_alloc(own, &sc.act, (size_t)T * sh.Ip * es, acct));
    FL_TRY(dev_alloc(own, (void **)&sc.ids, (size_t)T * 4, acct));
    return FL_OK;
}

// The prefill scratch of a shard grows geometrically and the set it replaces is freed: every forward() ends with a
// stream synchronisation, so under the model mutex the old buffers are idle (a long-running server that sees longer
// and longer prompts would otherwise pile up one dead set per new maximum, ~180 KB per token for Mistral-7B).
static int grow_prefill_scratch(Model *m, Shard &sh, int64_t T) {
    const int64_t chunk_max = tune(TK_PREFILL_CHUNK);
    int64_t cap = std::max<int64_t>(T, std::min<int64_t>(chunk_max, sh.pre.cap_T + sh.pre.cap_T / 2));
    cap = std::min<int64_t>(std::max<int64_t>(T, chunk_max), (cap + 127) / 128 * 128);
    FL_HIP(hipStreamSynchronize(sh.stream));
    if (sh.comm_stream) FL_HIP(hipStreamSynchronize(sh.comm_stream));
    for (void *p : sh.pre_allocs) (void)hipFree(p);
    sh.pre_allocs.clear();
    m->hbm_bytes -= sh.pre_bytes;
    sh.pre = Scratch{};
    const int64_t before = m->hbm_bytes;
    int rc = alloc_scratch(m, sh, sh.pre, cap, &sh.pre_allocs);
    if (rc != FL_OK) {                                   // leave the shard without a prefill scratch rather than with half of one
        for (void *p : sh.pre_allocs) (void)hipFree(p);
        sh.pre_allocs.clear(); sh.pre = Scratch{}; sh.pre_bytes = 0; m->hbm_bytes = before;
        return rc;
    }
    sh.pre_bytes = m->hbm_bytes - before;
    return FL_OK;
}

int model_create(const fl_config *cfg, const fl_tensor *tensors, size_t n, int compute_dtype,
                 const fl_parallel *par, Model **out) {
    if (!out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null out pointer");
    if (!tensors && n) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null tensors");
    if (compute_dtype != FL_DTYPE_BF16 && compute_dtype != FL_DTYPE_F32)
        FL_FAIL(FL_ERR_UNSUPPORTED, "compute dtype must be BF16 (reference default, main.rs:120) or F32");
    Dims D;
    FL_TRY(resolve_config(cfg, &D));
    if (D.h % 8) FL_FAIL(FL_ERR_UNSUPPORTED, "hidden_size must be a multiple of 8 (16-byte rows)");
    if (D.dm > 128) FL_FAIL(FL_ERR_UNSUPPORTED, "head_dim %lld not supported (even values up to 128)", (long long)D.dm);
    if (D.max_pos > (1 << 20)) D.max_pos = 1 << 20;
    debug_inject("model_create");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        FL_FAIL(FL_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU path");

    fl_parallel P{};
    if (par) P = *par;
    if (P.mode == FL_TP_NONE) { P.tp_size = 1; P.tp_rank = 0; }
    if (P.tp_size < 1) FL_FAIL(FL_ERR_BAD_ARGUMENT, "tp_size must be >= 1");
    const int tp = P.tp_size;
    if (D.H % tp || D.Hkv % tp || D.inter % tp)
        FL_FAIL(FL_ERR_UNSUPPORTED, "tp=%d must divide heads %lld, kv heads %lld and intermediate %lld", tp,
                (long long)D.H, (long long)D.Hkv, (long long)D.inter);

    std::unique_ptr<Model> m(new Model());
    m->D = D; m->dtype = compute_dtype; m->tp = tp; m->tp_mode = P.mode;
    m->vocab_parallel = tp > 1 && D.V % tp == 0;
    m->cfg_resolved = *cfg;
    m->cfg_resolved.num_key_value_heads = D.Hkv; m->cfg_resolved.rope_theta = D.theta;
    m->cfg_resolved.max_position_embeddings = D.max_pos; m->cfg_resolved.sliding_window = D.window;
    m->use_graph = tune(TK_GRAPH) != 0;                 // (-1 = automatic: on)
    m->fused_decode = tune(TK_FUSED) != 0 && gemv_norm_supported(compute_dtype, 1, D.h);

    auto dev_of = [&](int i) -> int { return (P.device_ids && i < P.n_device_ids) ? P.device_ids[i] : i; };
    int nlocal = 1;
    if (P.mode == FL_TP_SINGLE_PROCESS || P.mode == FL_TP_EMULATED) nlocal = tp;
    m->shards.resize(nlocal);
    for (int i = 0; i < nlocal; i++) {
        Shard &sh = m->shards[i];
        switch (P.mode) {
            case FL_TP_NONE: sh.rank = 0; sh.device = P.device_ids && P.n_device_ids > 0 ? P.device_ids[0] : 0; break;
            case FL_TP_SINGLE_PROCESS: sh.rank = i; sh.device = dev_of(i); break;
            case FL_TP_MULTI_PROCESS: sh.rank = P.tp_rank; sh.device = P.device_ids && P.n_device_ids > 0 ? P.device_ids[0] : 0; break;
            case FL_TP_EMULATED: sh.rank = i; sh.device = P.device_ids && P.n_device_ids > 0 ? P.device_ids[0] : 0; break;
            default: FL_FAIL(FL_ERR_BAD_ARGUMENT, "unknown tp mode %d", P.mode);
        }
        if (sh.rank < 0 || sh.rank >= tp) FL_FAIL(FL_ERR_BAD_ARGUMENT, "tp_rank %d out of range", sh.rank);
        if (sh.device < 0 || sh.device >= ndev) FL_FAIL(FL_ERR_NO_DEVICE, "device %d not present (%d visible)", sh.device, ndev);
        sh.Hs = D.H / tp; sh.Hkvs = D.Hkv / tp; sh.Is = D.inter / tp; sh.Ip = (sh.Is + 15) / 16 * 16;
        sh.Vs = m->vocab_parallel ? D.V / tp : D.V; sh.v0 = m->vocab_parallel ? sh.Vs * sh.rank : 0;
        FL_HIP(hipSetDevice(sh.device));
        if (P.mode == FL_TP_EMULATED && i > 0) sh.stream = m->shards[0].stream;     // one stream: sequential
        else FL_HIP(hipStreamCreateWithFlags(&sh.stream, hipStreamNonBlocking));
    }
    {   // is it a gfx950?
        hipDeviceProp_t prop;
        FL_HIP(hipGetDeviceProperties(&prop, m->shards[0].device));
        if (!strstr(prop.gcnArchName, "gfx950") && !tune(TK_ALLOW_ANY_ARCH))
            FL_FAIL(FL_ERR_NO_DEVICE, "device is %s; this library is built for gfx950 only", prop.gcnArchName);
    }

    Builder B; B.m = m.get();
    bool device_sources = false;
    for (size_t i = 0; i < n; i++) {
        if (!tensors[i].name) FL_FAIL(FL_ERR_BAD_ARGUMENT, "tensor %zu has no name", i);
        B.map[tensors[i].name] = &tensors[i];
        device_sources = device_sources || tensors[i].device >= 0;
    }
    if (device_sources) {
        // source tensors already in HBM may still be in flight on the caller's streams (a framework's generator or
        // loader); the conversion kernels run on this model's own streams, so wait for the devices first
        for (auto &sh : m->shards) { FL_HIP(hipSetDevice(sh.device)); FL_HIP(hipDeviceSynchronize()); }
    }
    FL_TRY(build_weights(B));
    FL_TRY(build_rope(m.get()));
    for (auto &sh : m->shards) {
        FL_HIP(hipSetDevice(sh.device));
        FL_TRY(alloc_scratch(m.get(), sh, sh.dec, 1));
        FL_TRY(dev_alloc(sh.allocs, (void **)&sh.logits_local, (size_t)sh.Vs * 4, &m->hbm_bytes));
        FL_TRY(dev_alloc(sh.allocs, (void **)&sh.logits_full, (size_t)D.V * 4, &m->hbm_bytes));
        FL_TRY(dev_alloc(sh.allocs, (void **)&sh.amax, sizeof(ArgmaxCand) * kMaxArgmaxCand, &m->hbm_bytes));
        // the persistent decode engine's granule edges and tag epoch (k_engine.hip); tags never repeat, so they are zeroed once
        if (compute_dtype == FL_DTYPE_BF16 && engine_shape_ok(D.h, sh.Hs * D.d, sh.Ip, sh.Vs) && D.L <= 63) {
            const size_t ne[3] = {(size_t)D.h, (size_t)sh.Ip / 2, (size_t)D.h};
            for (int e = 0; e < 3; e++) {
                FL_TRY(dev_alloc(sh.allocs, (void **)&sh.eng_edge[e], ne[e] * 8, &m->hbm_bytes));
                FL_HIP(hipMemsetAsync(sh.eng_edge[e], 0, ne[e] * 8, sh.stream));
            }
            FL_TRY(dev_alloc(sh.allocs, (void **)&sh.eng_epoch, 16, &m->hbm_bytes));
            FL_HIP(hipMemsetAsync(sh.eng_epoch, 0, 16, sh.stream));
        }
    }
    FL_HIP(hipSetDevice(m->shards[0].device));
    FL_HIP(hipHostMalloc((void **)&m->host_logits, (size_t)D.V * 4, hipHostMallocDefault));
    FL_HIP(hipHostMalloc((void **)&m->host_tokens, kOutTokensCap * 4, hipHostMallocDefault));
    FL_HIP(hipHostMalloc((void **)&m->host_state, sizeof(StepState), hipHostMallocDefault));
#ifdef FL_EXPERIMENTAL
    m->engine = tune(TK_ENGINE);                  // persistent decode engine (k_engine.hip): 1 = wherever it runs (opt-in: it measured slower)
    m->fuse_oproj = tune(TK_FUSE_OPROJ);          // 0.0-1.5 % at best (profiles/r02/README.md): off unless asked for; -1 = where it pays most
#else
    m->engine = 0; m->fuse_oproj = 0;             // measured losers live in the EXPERIMENTAL build only (Makefile)
#endif

    // communicators
    if (tp > 1 && P.mode == FL_TP_SINGLE_PROCESS) {
        // One process drives all tp GPUs (the reference's process model).  The inboxes of the one-shot collectives
        // are then plain peer pointers -- no IPC -- and because those collectives synchronise through memory, every
        // shard's decode step is an independent hipGraph on its own stream.  RCCL (group calls) carries the large
        // prefill collectives; it refuses two ranks on one device, so a group with repeated device ids (a one-GPU
        // rehearsal) runs everything one-shot.
        if (tp > FL_MAX_TP) FL_FAIL(FL_ERR_UNSUPPORTED, "tp_size %d > %d", tp, FL_MAX_TP);
        std::vector<int> devs; for (auto &sh : m->shards) devs.push_back(sh.device);
        bool distinct = true;
        for (int i = 0; i < tp; i++) for (int j = 0; j < i; j++) distinct = distinct && devs[i] != devs[j];
        if (distinct) {
            std::vector<ncclComm_t> comms(tp);
            FL_NCCL(ncclCommInitAll(comms.data(), tp, devs.data()));
            for (int i = 0; i < tp; i++) m->shards[i].comm = comms[i];
        }
        if (tune(TK_ONESHOT) || !distinct) {
            bool ok = true;
            for (int i = 0; i < tp && ok; i++) ok = comm_alloc(m.get(), m->shards[i]) == FL_OK;
            for (int i = 0; i < tp && ok; i++) {
                (void)hipSetDevice(devs[i]);
                for (int j = 0; j < tp && ok; j++) {
                    if (devs[j] == devs[i]) continue;
                    const hipError_t e = hipDeviceEnablePeerAccess(devs[j], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
                    (void)hipGetLastError();
                }
            }
            if (ok) {
                for (int i = 0; i < tp; i++) {
                    for (int r = 0; r < tp; r++) comm_set_entry(m->shards[i].pc, r, m->shards[r].pc.local);
                    m->shards[i].pc.connected = true;
                    m->shards[i].pc.shares_device = !distinct;
                    FL_TRY(comm_ll_publish(m.get(), m->shards[i]));
                }
            } else if (!distinct) {
                FL_FAIL(FL_ERR_RCCL, "cannot connect the shards of a single-device tensor-parallel group");
            }
        }
    } else if (tp > 1 && P.mode == FL_TP_MULTI_PROCESS) {
        // Small collectives (decode) go over peer-mapped inboxes; RCCL carries the large prefill ones.
        // Without a unique_id there is no RCCL communicator: the host must connect the inboxes itself
        // (fl_comm_ipc_export / fl_comm_ipc_connect) and every collective takes the one-shot path.
        if (tp > FL_MAX_TP) FL_FAIL(FL_ERR_UNSUPPORTED, "tp_size %d > %d", tp, FL_MAX_TP);
        FL_TRY(comm_alloc(m.get(), m->shards[0]));
        if (P.unique_id) {
            ncclUniqueId id; memcpy(&id, P.unique_id, sizeof id);
            FL_HIP(hipSetDevice(m->shards[0].device));
            FL_NCCL(ncclCommInitRank(&m->shards[0].comm, tp, id, P.tp_rank));
            if (tune(TK_ONESHOT)) FL_TRY(comm_bootstrap_over_rccl(m.get()));
        } else if (tune(TK_DEBUG_TP_LOOPBACK)) {
            FL_TRY(comm_connect_loopback(m.get()));
        }
    } else if (tp > 1 && P.mode == FL_TP_EMULATED) {
        FL_HIP(hipMalloc((void **)&m->emu_ptrs, sizeof(float *) * tp * 2));
    } else if (tp > 1) {
        FL_FAIL(FL_ERR_BAD_ARGUMENT, "tp_size %d needs a tensor-parallel mode", tp);
    } else if (tune(TK_DEBUG_RCCL_SELF)) {
        // single-GPU rehearsal of the RCCL plumbing: a 1-rank communicator whose all-reduce is the
        // identity, issued at the two real call sites (after o_proj and down_proj) on the compute stream
        ncclUniqueId id;
        FL_NCCL(ncclGetUniqueId(&id));
        FL_HIP(hipSetDevice(m->shards[0].device));
        FL_NCCL(ncclCommInitRank(&m->shards[0].comm, 1, id, 0));
        m->use_graph = tune(TK_GRAPH) > 0;              // eager unless graph capture of RCCL is asked for
        if (tune(TK_ONESHOT)) {                 // ... and of the inbox bootstrap: a group of one
            FL_TRY(comm_alloc(m.get(), m->shards[0]));
            FL_TRY(comm_bootstrap_over_rccl(m.get()));
        }
    }
    *out = m.release();
    return FL_OK;
}


// ------------------------------------------------------------------------------- cache
int cache_create(Model *m, size_t max_seq, Cache **out) {
    if (!m || !out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
    if (max_seq == 0) FL_FAIL(FL_ERR_BAD_ARGUMENT, "max_seq must be > 0");
    debug_inject("cache_create");
    const Dims &D = m->D;
    std::unique_ptr<Cache> c(new Cache());
    c->m = m; c->max_seq = max_seq; c->len = 0;
    // Under the model mutex and on the model's stream: another thread may be capturing its decode graph
    // on that stream, and a legacy-stream memset would try to join the capture.
    std::lock_guard<std::mutex> lock(m->mu);
    c->seq_alloc = (max_seq + 31) / 32 * 32;
    c->v_transposed = tune(TK_ATTN_MFMA) != 0 && attn_mfma_supported(m->dtype, m->shards[0].Hs, m->shards[0].Hkvs, D.d);
    // decode attention splits S so that the K/V stream of one kv head is spread over many CUs
    // (~64 cached positions per 4-wave workgroup at full length); partials are combined in-launch
    // MFMA kernel: 128 keys (four 32-key wave steps) per workgroup; VALU kernel: 64.  One CU pulls only
    // ~25-50 GB/s from HBM, so a kv head's K/V stream must be spread over many CUs -- except when it is
    // small (<= 96 KB per head): then one wide workgroup per head with no cross-workgroup combine wins.
    // Long caches: 256 keys (two steps per wave) -- fewer workgroups and half the partials to combine.  A/B inside one process
    // (tools/decode_ab.py, FL_ATTN_NSPLIT, tokens/s): Qwen2-7B at S = 4100 34 / 24 / 17 / 12 splits 361.3 / 363.1 / 364.9 / 359.6,
    // Mistral-7B at S = 4100 34 / 17 splits 340.5 / 347.1, at S = 2100 18 / 9 splits 358.0 / 356.9, at S = 530 7 / 4 370.4 / 364.5.
    // (A workgroup never takes fewer than 128 keys, so the split count of a large cache costs a short sequence nothing.)
    const int64_t keys_per_wg = c->v_transposed ? (max_seq > 2560 ? 256 : 128) : 64;
    int64_t ns = (int64_t)((max_seq + keys_per_wg - 1) / keys_per_wg);
    if (c->v_transposed && max_seq * (size_t)D.d * 4 <= 96 * 1024) ns = 1;
    c->nsplit = (int)std::max<int64_t>(1, std::min<int64_t>(ns, c->v_transposed ? 48 : 64));   // (measured at S = 8192 / 16384: 32..48 splits 17.4 / 24.0 us, 64: 18.7 / 25.4)
    if (tune(TK_ATTN_NSPLIT) >= 0) c->nsplit = tune(TK_ATTN_NSPLIT);
    {   // decode attention + o_proj in one launch when W_o's per-CU slice fits in LDS next to the attention state
        hipDeviceProp_t prop;
        FL_HIP(hipGetDeviceProperties(&prop, m->shards[0].device));
        // the fused launch's attention workgroups take 32 keys per wave and step; of their eight waves as few take keys as
        // keeps the splits few enough (a CU pulls only ~40 GB/s, so a split's K/V should stay small -- but every split
        // is a workgroup that holds no rows of W_o, and the others' LDS is full at ~156 rows)
        const int cus = prop.multiProcessorCount;
        const int w0 = tune(TK_AO_WAVES);
        c->fuse_oproj = false;
        for (int aw = w0 > 0 ? std::min(w0, 8) : 4; aw <= (w0 > 0 ? std::min(w0, 8) : 8) && !c->fuse_oproj; aw++) {
            const int ns8 = (int)std::max<int64_t>(1, (int64_t)((max_seq + 32 * aw - 1) / (32 * aw)));
            int nb = 0, ra = 0, ro = 0; size_t lds;
            const bool fits = m->fused_decode && c->v_transposed && m->tp == 1 && m->shards.size() == 1 && !m->shards[0].comm &&
                              attn_oproj_plan(m->shards[0].Hs, m->shards[0].Hkvs, D.d, D.h, ns8, cus, &nb, &ra, &ro, &lds);
            // where it pays (profiles/r02/README.md): the workgroups of a kv head are the 32 of one XCD (8 kv heads on 256 CUs)
            // and the attention workgroups own next to no rows, i.e. short contexts of Mistral-like shapes; TinyLlama / Qwen2
            // (4 kv heads: groups of 64 over two XCDs) and long contexts are faster as two launches
            const bool pays = fits && cus / m->shards[0].Hkvs <= 32 && ra <= 16;
            c->fuse_oproj = m->fuse_oproj > 0 ? fits : (m->fuse_oproj < 0 ? pays : false);
            c->ao_nsplit = ns8; c->ao_waves = aw;
        }
    }
    // short caches: attention replicated in every workgroup of the o_proj launch (k_attn_rep.hip): one launch and one dependent
    // step fewer per layer; FL_ATTN_REP=0 keeps the two launches
    c->rep_attn = tune(TK_ATTN_REP) != 0 && m->fused_decode && c->v_transposed && !c->fuse_oproj && m->dtype == FL_DTYPE_BF16 &&
                  attn_oproj_rep_supported(m->shards[0].Hs, m->shards[0].Hkvs, D.d, D.h, (int64_t)c->seq_alloc, tune(TK_ATTN_REP) == 2);
    c->shards.resize(m->shards.size());
    for (size_t i = 0; i < m->shards.size(); i++) {
        Shard &sh = m->shards[i]; CacheShard &cs = c->shards[i];
        FL_HIP(hipSetDevice(sh.device));
        const size_t kvb = (size_t)D.L * sh.Hkvs * c->seq_alloc * D.d * m->esize();
        FL_TRY(dev_alloc(cs.allocs, &cs.k, kvb, nullptr));
        FL_TRY(dev_alloc(cs.allocs, &cs.v, kvb, nullptr));
        // finite contents everywhere: the MFMA kernel multiplies masked keys' values by p = 0
        FL_HIP(hipMemsetAsync(cs.k, 0, kvb, sh.stream));
        FL_HIP(hipMemsetAsync(cs.v, 0, kvb, sh.stream));
        FL_TRY(dev_alloc(cs.allocs, (void **)&cs.st, sizeof(StepState), nullptr));
        FL_TRY(dev_alloc(cs.allocs, (void **)&cs.out_tokens, kOutTokensCap * 4, nullptr));
        FL_TRY(dev_alloc(cs.allocs, (void **)&cs.part_m, (size_t)sh.Hs * c->nsplit * 4, nullptr));
        FL_TRY(dev_alloc(cs.allocs, (void **)&cs.part_l, (size_t)sh.Hs * c->nsplit * 4, nullptr));
        FL_TRY(dev_alloc(cs.allocs, (void **)&cs.part_o, (size_t)sh.Hs * c->nsplit * D.d * 4, nullptr));
        FL_TRY(dev_alloc(cs.allocs, (void **)&cs.counters, (size_t)sh.Hs * 4, nullptr));
        FL_HIP(hipMemsetAsync(cs.counters, 0, (size_t)sh.Hs * 4, sh.stream));
        FL_TRY(dev_alloc(cs.allocs, (void **)&cs.ss, sizeof(SampleState), nullptr));
        FL_HIP(hipMemsetAsync(cs.ss, 0, sizeof(SampleState), sh.stream));
        FL_TRY(dev_alloc(cs.allocs, (void **)&cs.sel_scratch, (size_t)D.V * 4, nullptr));
        if (c->fuse_oproj) {
            FL_TRY(dev_alloc(cs.allocs, (void **)&cs.ao_part, (size_t)sh.Hs * c->ao_nsplit * (D.d + 4) * 4, nullptr));
        }
        FL_TRY(dev_alloc(cs.allocs, (void **)&cs.heads_done, (size_t)D.L * sh.Hkvs * 4, nullptr));
        FL_HIP(hipMemsetAsync(cs.heads_done, 0, (size_t)D.L * sh.Hkvs * 4, sh.stream));
        FL_HIP(hipMemsetAsync(cs.st, 0, sizeof(StepState), sh.stream));
        FL_HIP(hipStreamSynchronize(sh.stream));
    }
    m->refs.fetch_add(1);
    *out = c.release();
    return FL_OK;
}

// ------------------------------------------------------------------------------- forward
__global__ void set_state_kernel(StepState *st, uint32_t token, uint32_t pos, uint32_t len, uint32_t call0, uint32_t step, int32_t eos,
                                 unsigned *heads_done, int n_layers, SampleState *ss, SampleState ss_new, int ss_set) {
    if (threadIdx.x == 0) {
        st->token = token; st->pos = pos; st->len = len; st->call0 = call0; st->step = step; st->eos = eos; st->done = 0; st->error = 0;
        if (ss_set) *ss = ss_new;
    }
    for (int l = threadIdx.x; l < n_layers; l += blockDim.x) heads_done[l] = 0;       // targets restart with step
}

// LogitsProcessor::new(seed, Some(temperature), None) (mod.rs:373-374): ArgMax below 1e-7, else the
// StdRng stream of rand 0.8 -- ChaCha12 keyed by SeedableRng::seed_from_u64 (PCG32 XSH-RR expansion of
// the u64 into 8 little-endian key words) -- positioned after `draws_done` u32 words.
SampleState make_sampler(const fl_sampling *sp) {
    SampleState s{};
    if (!sp || !(sp->temperature >= 1e-7)) return s;
    s.on = tune(TK_SAMPLE_WALK) ? 2 : 1;       // 2: plain one-lane walk instead of ordered_sum (cross-check)
    s.inv_temp = (float)(1.0 / sp->temperature);
    uint64_t state = sp->seed;
    for (int i = 0; i < 8; i++) {
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27), rot = (uint32_t)(state >> 59);
        s.key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    s.draw_lo = (uint32_t)sp->draws_done; s.draw_hi = (uint32_t)(sp->draws_done >> 32);
    return s;
}

Launcher make_launcher(Model *m, Shard &sh) {
    Launcher L; L.stream = sh.stream; L.prof = m->profiling ? &m->prof : nullptr; L.tp = m->tp; return L;
}

// all-reduce(sum) of each local shard's `delta` [count] fp32 (after o_proj / down_proj rows)
static int all_reduce_delta(Model *m, bool pre, int64_t count) {
    if (m->tp == 1 && !m->shards[0].comm) return FL_OK;
    if (m->tp_mode == FL_TP_EMULATED) {
        std::vector<float *> ptrs;
        for (auto &sh : m->shards) ptrs.push_back(pre ? sh.pre.delta : sh.dec.delta);
        Shard &s0 = m->shards[0];
        FL_HIP(hipSetDevice(s0.device));
        float **tab = m->emu_ptrs + (pre ? m->tp : 0);
        FL_HIP(hipMemcpyAsync(tab, ptrs.data(), sizeof(float *) * m->tp, hipMemcpyHostToDevice, s0.stream));
        FL_HIP(hipStreamSynchronize(s0.stream));          // ptrs is a stack vector
        Launcher L = make_launcher(m, s0);
        return launch_reduce_shards(L, tab, m->tp, count);
    }
    if (m->tp_mode == FL_TP_SINGLE_PROCESS) {
        Shard &s0 = m->shards[0];
        if (s0.pc.connected && (count <= s0.pc.nmax || !s0.comm)) {
            for (auto &sh : m->shards) {
                FL_HIP(hipSetDevice(sh.device));
                float *buf = pre ? sh.pre.delta : sh.dec.delta;
                FL_TRY(oneshot(m, sh, false, buf, buf, count, 0));
            }
            return FL_OK;
        }
    } else if (m->tp_mode == FL_TP_MULTI_PROCESS || m->shards[0].pc.connected) {
        Shard &sh = m->shards[0];
        float *buf = pre ? sh.pre.delta : sh.dec.delta;
        if (sh.pc.connected && (count <= sh.pc.nmax || !sh.comm)) return oneshot(m, sh, false, buf, buf, count, 0);
        if (!sh.comm) FL_FAIL(FL_ERR_RCCL, "tensor-parallel group is not connected: call fl_comm_ipc_connect first");
    }
    FL_NCCL(ncclGroupStart());
    for (auto &sh : m->shards) {
        float *buf = pre ? sh.pre.delta : sh.dec.delta;
        FL_NCCL(ncclAllReduce(buf, buf, (size_t)count, ncclFloat, ncclSum, sh.comm, sh.stream));
    }
    FL_NCCL(ncclGroupEnd());
    return FL_OK;
}

// where a shard's lm_head launch writes: its slice of a sharded vocabulary (gathered below), else the full logits at once
// (a 128 KB device-to-device copy per decode step used to stand here)
static float *lm_head_out(Model *m, Shard &sh) { return m->vocab_parallel ? sh.logits_local : sh.logits_full; }

static int gather_logits(Model *m) {
    if (!m->vocab_parallel) return FL_OK;
    if (m->tp_mode == FL_TP_EMULATED) {
        Shard &s0 = m->shards[0];
        FL_HIP(hipSetDevice(s0.device));
        for (auto &dst : m->shards)
            for (auto &src : m->shards)
                FL_HIP(hipMemcpyAsync(dst.logits_full + src.v0, src.logits_local, (size_t)src.Vs * 4, hipMemcpyDeviceToDevice, s0.stream));
        return FL_OK;
    }
    if (m->tp_mode == FL_TP_SINGLE_PROCESS && m->shards[0].pc.connected && (m->shards[0].Vs <= m->shards[0].pc.nmax || !m->shards[0].comm)) {
        for (auto &sh : m->shards) {
            FL_HIP(hipSetDevice(sh.device));
            FL_TRY(oneshot(m, sh, true, sh.logits_local, sh.logits_full, sh.Vs, sh.Vs));
        }
        return FL_OK;
    }
    if (m->tp_mode == FL_TP_MULTI_PROCESS) {
        Shard &sh = m->shards[0];
        if (sh.pc.connected && (sh.Vs <= sh.pc.nmax || !sh.comm)) return oneshot(m, sh, true, sh.logits_local, sh.logits_full, sh.Vs, sh.Vs);
        if (!sh.comm) FL_FAIL(FL_ERR_RCCL, "tensor-parallel group is not connected: call fl_comm_ipc_connect first");
    }
    FL_NCCL(ncclGroupStart());
    for (auto &sh : m->shards)
        FL_NCCL(ncclAllGather(sh.logits_local, sh.logits_full, (size_t)sh.Vs, ncclFloat, sh.comm, sh.stream));
    FL_NCCL(ncclGroupEnd());
    return FL_OK;
}

static bool fused_all_reduce(Model *m, Cache *c) { return !c->fuse_oproj && fused_all_reduce_ready(m); }

// The decode step on the persistent engine (k_engine.hip): per layer the attention launch and ONE launch that chains
// o_proj -> gate/up -> down_proj -> the next layer's QKV projection (the last layer: lm_head) -- 2 L + 1 launches instead of 5 L + 1.
// One shard, no tensor parallelism (stage 1), bf16, MFMA attention; FL_ENGINE=1 turns it on (default off: see below).
static bool engine_usable(Model *m, Cache *c) {
    const int want = m->engine;
    if (want == 0 || m->shards.size() != 1 || m->tp != 1 || m->dtype != FL_DTYPE_BF16 || !m->fused_decode || c->fuse_oproj || !c->v_transposed) return false;
    Shard &sh = m->shards[0];
    if (!sh.eng_epoch || sh.pc.shares_device || m->D.L < 1) return false;
    if (!gemv_norm_supported(m->dtype, (sh.Hs + 2 * sh.Hkvs) * m->D.d, m->D.h)) return false;
    // opt-in: measured 5 % SLOWER than the five launches it replaces on TinyLlama-1.1B (24.4 us against 23.2 us per layer for
    // o_proj .. QKV; profiles/r03/README.md has the in-kernel stamps: every chip-wide edge costs 2.2-3 us, a kernel boundary
    // + ramp 2.5-3, and the prefetch across an edge only moves the pipeline's fill bubble behind the barrier)
    return want == 1;
}

static int enqueue_decode_engine(Model *m, Cache *c, int64_t len_hint) {
    const Dims &D = m->D;
    const int dt = m->dtype;
    Shard &sh = m->shards[0]; Scratch &sc = sh.dec; CacheShard &cs = c->shards[0];
    FL_HIP(hipSetDevice(sh.device));
    Launcher L = make_launcher(m, sh);
    const long long timeout_ticks = (long long)tune(TK_ENGINE_TIMEOUT_MS) * 100000ll;
    auto kv_of = [&](int64_t l, void **kc, void **vc) {
        const size_t kv_layer = (size_t)l * sh.Hkvs * c->seq_alloc * D.d * m->esize();
        *kc = (char *)cs.k + kv_layer; *vc = (char *)cs.v + kv_layer;
    };
    for (int64_t l = 0; l < D.L; l++) {
        LayerW &ly = sh.layers[l];
        void *kc, *vc;
        kv_of(l, &kc, &vc);
        if (l == 0) {                                  // the first QKV projection reads the token's embedding row: the launch of k_gemv.hip
            GemvArgs a;
            a.W = ly.wqkv; a.bias = ly.bqkv; a.N = (int)((sh.Hs + 2 * sh.Hkvs) * D.d); a.K = (int)D.h;
            a.epi = EPI_QKV_ROPE; a.pro = PRO_NORM; a.norm_w = ly.ln1; a.eps = D.eps; a.st = cs.st;
            a.embed = sh.embed; a.x_out = sc.x_res2;
            a.cos_tab = sh.cos_tab; a.sin_tab = sh.sin_tab; a.q_out = sc.q; a.k_cache = kc; a.v_cache = vc;
            a.H = (int)sh.Hs; a.Hkv = (int)sh.Hkvs; a.d = (int)D.d; a.max_seq = (int)c->seq_alloc; a.max_pos = (int)D.max_pos;
            a.v_ld = (int)c->seq_alloc;
            FL_TRY(launch_gemv(L, dt, a));
        }
        AttnScratch as{cs.part_m, cs.part_l, cs.part_o, cs.counters, c->nsplit, len_hint + 1};
        FL_TRY(launch_attn_decode_mfma(L, sc.q, kc, vc, cs.st, sc.ao, as, sh.Hs, sh.Hkvs, D.d, (int64_t)c->seq_alloc, D.scale));
        const bool last = l + 1 == D.L;
        float *res_in = (l & 1) ? sc.x_res : sc.x_res2, *res_out = (l & 1) ? sc.x_res2 : sc.x_res;
        EngArgs e;
        e.nops = 4; e.h = (int)D.h; e.x_res_in = res_in; e.eps = D.eps; e.st = cs.st; e.st_rw = cs.st; e.epoch = sh.eng_epoch;
        e.timeout_ticks = timeout_ticks;
        const int t0 = (int)(4 * l);
        EngOp &o0 = e.op[0], &o1 = e.op[1], &o2 = e.op[2], &o3 = e.op[3];
        o0.W = ly.wo; o0.N = (int)D.h; o0.K = (int)(sh.Hs * D.d); o0.in = ENG_IN_X; o0.x = sc.ao; o0.out = ENG_OUT_EDGE_F32; o0.out_edge = sh.eng_edge[0]; o0.tag_out = t0 + 1;
        o1.W = ly.wgu; o1.N = (int)(2 * sh.Ip); o1.K = (int)D.h; o1.in = ENG_IN_NORM; o1.norm_w = ly.ln2; o1.in_edge = sh.eng_edge[0]; o1.tag_in = t0 + 1;
        o1.out = ENG_OUT_EDGE_ACT; o1.out_edge = sh.eng_edge[1]; o1.tag_out = t0 + 2;
        o2.W = ly.wd; o2.N = (int)D.h; o2.K = (int)sh.Ip; o2.in = ENG_IN_ACT; o2.in_edge = sh.eng_edge[1]; o2.tag_in = t0 + 2;
        o2.out = ENG_OUT_EDGE_F32; o2.out_edge = sh.eng_edge[2]; o2.tag_out = t0 + 3;
        o3.K = (int)D.h; o3.in = ENG_IN_NORM; o3.in_edge = sh.eng_edge[2]; o3.tag_in = t0 + 3;
        if (!last) {
            LayerW &nx = sh.layers[l + 1];
            void *kn, *vn;
            kv_of(l + 1, &kn, &vn);
            o3.W = nx.wqkv; o3.N = (int)((sh.Hs + 2 * sh.Hkvs) * D.d); o3.norm_w = nx.ln1; o3.bias = nx.bqkv; o3.out = ENG_OUT_QKV; o3.res_out = res_out;
            e.cos_tab = sh.cos_tab; e.sin_tab = sh.sin_tab; e.q_out = sc.q; e.k_cache = kn; e.v_cache = vn;
            e.H = (int)sh.Hs; e.Hkv = (int)sh.Hkvs; e.d = (int)D.d; e.max_seq = (int)c->seq_alloc; e.max_pos = (int)D.max_pos; e.v_ld = (int)c->seq_alloc;
        } else {
            o3.W = sh.lm_head; o3.N = (int)sh.Vs; o3.norm_w = sh.norm; o3.out = ENG_OUT_LOGITS; o3.dst = sh.logits_full;
            if (tune(TK_ARGMAX_FUSED)) { e.amax = sh.amax; sh.amax_valid = true; }
        }
        FL_TRY(launch_engine(L, e));
    }
    return FL_OK;
}

static int enqueue_decode_fused(Model *m, Cache *c, int64_t len_hint) {
    if (engine_usable(m, c)) return enqueue_decode_engine(m, c, len_hint);
    const Dims &D = m->D;
    const int dt = m->dtype;
    const size_t ns = m->shards.size();
    const bool far = fused_all_reduce(m, c);
    // out = sum over ranks of W[h, K] . x  -- the row-parallel projection with the exchange in its epilogue
    auto row_parallel = [&](Launcher &L, Shard &sh, const void *W, const void *x, int64_t K, float *out, int slot) -> int {
        GemvArgs a;
        a.W = W; a.x = x; a.out = out; a.N = (int)D.h; a.K = (int)K; a.epi = EPI_F32; a.pro = PRO_X;
        a.ll = sh.pc.ll_dev; a.ll_slot = slot;
        return launch_gemv(L, dt, a);
    };
    for (int64_t l = 0; l < D.L; l++) {
        for (size_t i = 0; i < ns; i++) {
            Shard &sh = m->shards[i]; Scratch &sc = sh.dec; CacheShard &cs = c->shards[i]; LayerW &ly = sh.layers[l];
            FL_HIP(hipSetDevice(sh.device));
            Launcher L = make_launcher(m, sh);
            const size_t kv_layer = (size_t)l * sh.Hkvs * c->seq_alloc * D.d * m->esize();
            void *kc = (char *)cs.k + kv_layer, *vc = (char *)cs.v + kv_layer;
            GemvArgs a;
            a.W = ly.wqkv; a.bias = ly.bqkv; a.N = (int)((sh.Hs + 2 * sh.Hkvs) * D.d); a.K = (int)D.h;
            a.epi = EPI_QKV_ROPE; a.pro = PRO_NORM; a.norm_w = ly.ln1; a.eps = D.eps; a.st = cs.st;
            if (l == 0) { a.embed = sh.embed; a.x_out = sc.x_res2; }
            else { a.x_in = sc.x_res; a.delta = sc.delta; a.x_out = sc.x_res2; }
            a.cos_tab = sh.cos_tab; a.sin_tab = sh.sin_tab; a.q_out = sc.q; a.k_cache = kc; a.v_cache = vc;
            a.H = (int)sh.Hs; a.Hkv = (int)sh.Hkvs; a.d = (int)D.d; a.max_seq = (int)c->seq_alloc; a.max_pos = (int)D.max_pos;
            a.v_ld = c->v_transposed ? (int)c->seq_alloc : 0;
            FL_TRY(launch_gemv(L, dt, a));
            AttnScratch as{cs.part_m, cs.part_l, cs.part_o, cs.counters, c->nsplit, len_hint + 1};
            as.prefetch = ly.wo; as.prefetch_bytes = D.h * sh.Hs * D.d * (int64_t)m->esize();   // o_proj's weights, while HBM idles under the attention
            as.prefetch_chunk = gemv_owner_chunk(dt, D.h, sh.Hs * D.d); as.prefetch_row = sh.Hs * D.d * (int64_t)m->esize();
            if (c->fuse_oproj) {
                FL_TRY(launch_attn_oproj(L, sc.q, kc, vc, cs.st, cs.st, cs.ao_part, cs.heads_done + l * sh.Hkvs, c->ao_nsplit, c->ao_waves, len_hint + 1, ly.wo,
                                         sc.delta, sh.Hs, sh.Hkvs, D.d, D.h, (int64_t)c->seq_alloc, D.scale));
            } else if (c->rep_attn) {
                AttnRepArgs ra;
                ra.q = sc.q; ra.kc = kc; ra.vT = vc; ra.st = cs.st; ra.Wo = ly.wo; ra.out = sc.delta;
                ra.H = (int)sh.Hs; ra.Hkv = (int)sh.Hkvs; ra.seq_alloc = (int)c->seq_alloc; ra.N = (int)D.h; ra.K = (int)(sh.Hs * D.d); ra.scale = D.scale;
                if (far) { ra.ll = sh.pc.ll_dev; ra.ll_slot = (int)(2 * l + 1); }
                FL_TRY(launch_attn_oproj_rep(L, ra));
            } else {
                if (c->v_transposed) FL_TRY(launch_attn_decode_mfma(L, sc.q, kc, vc, cs.st, sc.ao, as, sh.Hs, sh.Hkvs, D.d, (int64_t)c->seq_alloc, D.scale));
                else FL_TRY(launch_attn_decode(L, dt, sc.q, kc, vc, cs.st, sc.ao, as, sh.Hs, sh.Hkvs, D.d, (int64_t)c->seq_alloc, D.scale));
                if (far) FL_TRY(row_parallel(L, sh, ly.wo, sc.ao, sh.Hs * D.d, sc.delta, (int)(2 * l + 1)));
                else FL_TRY(launch_linear(L, dt, ly.wo, sc.ao, nullptr, sc.delta, 1, D.h, sh.Hs * D.d, EPI_F32));
            }
        }
        if (!far) FL_TRY(all_reduce_delta(m, false, D.h));
        for (size_t i = 0; i < ns; i++) {
            Shard &sh = m->shards[i]; Scratch &sc = sh.dec; LayerW &ly = sh.layers[l];
            FL_HIP(hipSetDevice(sh.device));
            Launcher L = make_launcher(m, sh);
            GemvArgs a;
            a.W = ly.wgu; a.out = sc.act; a.N = (int)(2 * sh.Ip); a.K = (int)D.h; a.epi = EPI_GATEUP; a.pro = PRO_NORM;
            a.x_in = sc.x_res2; a.delta = sc.delta; a.norm_w = ly.ln2; a.eps = D.eps; a.x_out = sc.x_res; a.st = c->shards[i].st;
            if (c->fuse_oproj) a.delta_nslab = (int)sh.Hkvs;           // one partial vector per kv head
            FL_TRY(launch_gemv(L, dt, a));
            if (far) FL_TRY(row_parallel(L, sh, ly.wd, sc.act, sh.Ip, sc.delta, (int)(2 * l + 2)));
            else FL_TRY(launch_linear(L, dt, ly.wd, sc.act, nullptr, sc.delta, 1, D.h, sh.Ip, EPI_F32));
        }
        if (!far) FL_TRY(all_reduce_delta(m, false, D.h));
    }
    for (size_t i = 0; i < ns; i++) {
        Shard &sh = m->shards[i]; Scratch &sc = sh.dec;
        FL_HIP(hipSetDevice(sh.device));
        Launcher L = make_launcher(m, sh);
        GemvArgs a;
        a.W = sh.lm_head; a.out = lm_head_out(m, sh); a.N = (int)sh.Vs; a.K = (int)D.h; a.epi = EPI_F32; a.pro = PRO_NORM;
        if (!m->vocab_parallel && tune(TK_ARGMAX_FUSED) && gemv_leaves_candidates(dt, a)) { a.amax = sh.amax; sh.amax_valid = true; }   // token selection reads one candidate per workgroup
        a.x_in = sc.x_res; a.delta = sc.delta; a.norm_w = sh.norm; a.eps = D.eps; a.st = c->shards[i].st;
        FL_TRY(launch_gemv(L, dt, a));
    }
    return gather_logits(m);
}

// all-reduce(sum) of rows [off, off + count) floats of every local shard's prefill delta, on the shards' comm streams
static int all_reduce_span_side(Model *m, size_t off, int64_t count) {
    const bool local = m->tp_mode == FL_TP_SINGLE_PROCESS;
    Shard &s0 = m->shards[0];
    if (s0.pc.connected && (count <= s0.pc.nmax || !s0.comm)) {
        for (auto &sh : m->shards) {
            FL_HIP(hipSetDevice(sh.device));
            FL_TRY(oneshot(m, sh, false, sh.pre.delta + off, sh.pre.delta + off, count, 0, sh.comm_stream));
        }
        return FL_OK;
    }
    if (!s0.comm) FL_FAIL(FL_ERR_RCCL, "tensor-parallel group is not connected: call fl_comm_ipc_connect first");
    if (local) FL_NCCL(ncclGroupStart());
    for (auto &sh : m->shards)
        FL_NCCL(ncclAllReduce(sh.pre.delta + off, sh.pre.delta + off, (size_t)count, ncclFloat, ncclSum, sh.comm, sh.comm_stream));
    if (local) FL_NCCL(ncclGroupEnd());
    return FL_OK;
}

// Tensor-parallel prefill with the all-reduces on a side stream (north_star: "RCCL all-reduce ... overlapped on a side
// HIP stream").  The T tokens are cut into two row chunks; every op between attention and the next attention is
// row-wise, so while chunk 0's o_proj output is being all-reduced, chunk 1's o_proj runs; while chunk 1's is reduced,
// chunk 0's norm / gate-up / down run; and so on into the next layer's norm + QKV.  Only RoPE / attention wait for both.
//   main:  ... attn(all) | o(c0) e0 | o(c1) e1 | wait h0: mlp(c0) f0 | wait h1: mlp(c1) f1 | wait g0: qkv(c0) | wait g1: qkv(c1) | rope, attn ...
//   comm:                 wait e0: AR(c0) h0 | wait e1: AR(c1) h1 | wait f0: AR(c0) g0 | wait f1: AR(c1) g1
static int enqueue_prefill_tp_overlap(Model *m, Cache *c, int64_t T) {
    const Dims &D = m->D;
    const int dt = m->dtype;
    const size_t es = m->esize();
    const int64_t T0 = std::min<int64_t>(T - 1, ((T / 2 + 255) / 256) * 256), T1 = T - T0;
    const int64_t rows[2] = {T0, T1}, row0[2] = {0, T0};
    enum { E0 = 0, E1, H0, H1, F0, F1, G0, G1 };
    for (auto &sh : m->shards) {
        FL_HIP(hipSetDevice(sh.device));
        if (!sh.comm_stream) FL_HIP(hipStreamCreateWithFlags(&sh.comm_stream, hipStreamNonBlocking));
        for (auto &e : sh.ev) if (!e) FL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        Launcher L = make_launcher(m, sh);
        FL_TRY(launch_embed(L, dt, sh.embed, sh.pre.ids, c->shards[&sh - &m->shards[0]].st, sh.pre.x_res, T, D.h));
    }
    auto each = [&](auto fn) -> int {
        for (size_t i = 0; i < m->shards.size(); i++) {
            Shard &sh = m->shards[i];
            FL_HIP(hipSetDevice(sh.device));
            FL_TRY(fn(sh, c->shards[i]));
        }
        return FL_OK;
    };
    auto rec = [&](int ev, bool side) { return each([&](Shard &sh, CacheShard &) -> int { FL_HIP(hipEventRecord(sh.ev[ev], side ? sh.comm_stream : sh.stream)); return FL_OK; }); };
    auto wait = [&](int ev, bool side) { return each([&](Shard &sh, CacheShard &) -> int { FL_HIP(hipStreamWaitEvent(side ? sh.comm_stream : sh.stream, sh.ev[ev], 0)); return FL_OK; }); };
    for (int64_t l = 0; l < D.L; l++) {
        for (int k = 0; k < 2; k++) {                                  // norm1 + QKV per chunk, as soon as its rows are reduced
            if (l > 0) FL_TRY(wait(G0 + k, false));
            FL_TRY(each([&](Shard &sh, CacheShard &) -> int {
                Scratch &sc = sh.pre; LayerW &ly = sh.layers[l]; Launcher L = make_launcher(m, sh);
                const int64_t nq = (sh.Hs + 2 * sh.Hkvs) * D.d;
                const size_t r = (size_t)row0[k];
                FL_TRY(launch_rmsnorm_add(L, dt, sc.x_res + r * D.h, l == 0 ? nullptr : sc.delta + r * D.h, ly.ln1, D.eps, (char *)sc.xn + r * D.h * es,
                                          sc.inv_rms + r, rows[k], D.h, 1, 0));
                return launch_linear(L, dt, ly.wqkv, (char *)sc.xn + r * D.h * es, ly.bqkv, sc.qkv + r * nq, rows[k], nq, D.h, EPI_F32, sc.inv_rms + r);
            }));
        }
        FL_TRY(each([&](Shard &sh, CacheShard &cs) -> int {
            Scratch &sc = sh.pre; Launcher L = make_launcher(m, sh);
            const size_t kv_layer = (size_t)l * sh.Hkvs * c->seq_alloc * D.d * es;
            void *kc = (char *)cs.k + kv_layer, *vc = (char *)cs.v + kv_layer;
            const int64_t sa = (int64_t)c->seq_alloc;
            FL_TRY(launch_rope_kv(L, dt, sc.qkv, cs.st, sh.cos_tab, sh.sin_tab, D.max_pos, sc.q, kc, vc, T, sh.Hs, sh.Hkvs, D.d, sa, c->v_transposed));
            if (c->v_transposed) return launch_attn_prefill_mfma(L, sc.q, kc, vc, cs.st, sc.ao, T, sh.Hs, sh.Hkvs, D.d, sa, D.scale, D.window);
            return launch_attn_prefill(L, dt, sc.q, kc, vc, cs.st, sc.ao, T, sh.Hs, sh.Hkvs, D.d, sa, D.scale, D.window);
        }));
        for (int k = 0; k < 2; k++) {                                  // o_proj per chunk; its all-reduce goes to the side stream
            FL_TRY(each([&](Shard &sh, CacheShard &) -> int {
                Scratch &sc = sh.pre; Launcher L = make_launcher(m, sh);
                const size_t r = (size_t)row0[k];
                return launch_linear(L, dt, sh.layers[l].wo, (char *)sc.ao + r * sh.Hs * D.d * es, nullptr, sc.delta + r * D.h, rows[k], D.h, sh.Hs * D.d, EPI_F32);
            }));
            FL_TRY(rec(E0 + k, false));
            FL_TRY(wait(E0 + k, true));
            FL_TRY(all_reduce_span_side(m, (size_t)row0[k] * D.h, rows[k] * D.h));
            FL_TRY(rec(H0 + k, true));
        }
        for (int k = 0; k < 2; k++) {                                  // MLP per chunk
            FL_TRY(wait(H0 + k, false));
            FL_TRY(each([&](Shard &sh, CacheShard &) -> int {
                Scratch &sc = sh.pre; LayerW &ly = sh.layers[l]; Launcher L = make_launcher(m, sh);
                const size_t r = (size_t)row0[k];
                FL_TRY(launch_rmsnorm_add(L, dt, sc.x_res + r * D.h, sc.delta + r * D.h, ly.ln2, D.eps, (char *)sc.xn + r * D.h * es, sc.inv_rms + r, rows[k], D.h, 1, 0));
                FL_TRY(launch_linear(L, dt, ly.wgu, (char *)sc.xn + r * D.h * es, nullptr, (char *)sc.act + r * sh.Ip * es, rows[k], 2 * sh.Ip, D.h, EPI_GATEUP, sc.inv_rms + r));
                return launch_linear(L, dt, ly.wd, (char *)sc.act + r * sh.Ip * es, nullptr, sc.delta + r * D.h, rows[k], D.h, sh.Ip, EPI_F32);
            }));
            FL_TRY(rec(F0 + k, false));
            FL_TRY(wait(F0 + k, true));
            FL_TRY(all_reduce_span_side(m, (size_t)row0[k] * D.h, rows[k] * D.h));
            FL_TRY(rec(G0 + k, true));
        }
    }
    FL_TRY(wait(G1, false));                                           // the last token lives in chunk 1
    FL_TRY(each([&](Shard &sh, CacheShard &) -> int {
        Scratch &sc = sh.pre; Launcher L = make_launcher(m, sh);
        float *xl = sc.x_res + (size_t)(T - 1) * D.h, *dl = sc.delta + (size_t)(T - 1) * D.h;
        void *xnl = (char *)sc.xn + (size_t)(T - 1) * D.h * es;
        FL_TRY(launch_rmsnorm_add(L, dt, xl, dl, sh.norm, D.eps, xnl, sc.inv_rms + (T - 1), 1, D.h, 1, 0));
        return launch_linear(L, dt, sh.lm_head, xnl, nullptr, lm_head_out(m, sh), 1, sh.Vs, D.h, EPI_F32, sc.inv_rms + (T - 1));
    }));
    FL_TRY(wait(G0, false));                                           // nothing of this call may still run on the side stream afterwards
    return gather_logits(m);
}

// Enqueue one forward over T tokens on every local shard.  The step state (pos, len, token) of the
// cache must already be set on the device.  ids_dev == null: the single token comes from the state.
static int enqueue_forward(Model *m, Cache *c, bool pre, int64_t T, bool ids_in_scratch, int64_t len_hint) {
    for (auto &sh : m->shards) sh.amax_valid = false;                  // (set by the path whose lm_head launch leaves ArgMax candidates)
    if (T == 1 && !pre && !ids_in_scratch && m->fused_decode) return enqueue_decode_fused(m, c, len_hint);
    if (pre && ids_in_scratch && m->tp > 1 && tune(TK_TP_OVERLAP) && T >= tune(TK_TP_OVERLAP_MIN_T) && !m->profiling &&
        (m->tp_mode == FL_TP_MULTI_PROCESS || m->tp_mode == FL_TP_SINGLE_PROCESS))
        return enqueue_prefill_tp_overlap(m, c, T);
    const Dims &D = m->D;
    const int dt = m->dtype;
    const size_t ns = m->shards.size();
    auto SC = [&](Shard &sh) -> Scratch & { return pre ? sh.pre : sh.dec; };
    for (size_t i = 0; i < ns; i++) {
        Shard &sh = m->shards[i]; Scratch &sc = SC(sh);
        FL_HIP(hipSetDevice(sh.device));
        Launcher L = make_launcher(m, sh);
        FL_TRY(launch_embed(L, dt, sh.embed, ids_in_scratch ? sc.ids : nullptr, c->shards[i].st, sc.x_res, T, D.h));
    }
    // split-K of the row-parallel GEMMs (o_proj, down_proj) only without tensor parallelism: the
    // all-reduce wants one summed buffer
    const int max_split = (m->tp == 1 && T > 1) ? ksplit_cap(T) : 1;
    const int64_t slab = T * D.h;
    int nslab = 1;                        // slabs the current delta consists of (same on every shard)
    // Long prompts on one GPU: where the 256x256 kernel takes o_proj / down_proj in one piece, its epilogue adds the residual,
    // writes the next norm's x * w and leaves partial sums of squares (EPI_RESID) -- no delta round trip, no rmsnorm_add launch.
    const bool resid_ok = ns == 1 && m->tp == 1 && !m->shards[0].comm && dt == FL_DTYPE_BF16 && T > 1 && SC(m->shards[0]).rs_part != nullptr;
    bool norm_done = false;               // xn / inv_rms for the upcoming norm were produced by the previous projection
    // consumer_takes_parts: the projection that follows takes its row scales (1/rms) straight from the partial sums (Launcher::rsp,
    // kernels.h) -- then there is no rms_finalize launch either; rs_lazy says so until that projection is launched
    bool rs_lazy = false;
    auto linear_resid = [&](Launcher &L, Shard &sh, Scratch &sc, const void *W, const void *x, int64_t K, const float *next_norm_w,
                            bool consumer_takes_parts) -> int {
        ResidEpi re;
        re.h = sc.x_res; re.w = next_norm_w; re.xn = sc.xn; re.part = sc.rs_part; re.np = gemm_resid_partials(D.h);
        FL_TRY(launch_gemm_resid(L, W, x, T, D.h, K, re));
        // (A/B, whole prefills: Mistral-7B 384 / 512 / 640 tokens 0.987 / 0.997 / 0.999, 4096 tokens 1.011: every workgroup of a long
        // prompt's grid sums 256 rows' partials again, the finalize launch does it once)
        rs_lazy = consumer_takes_parts && tune(TK_RS_LAZY) != 0 && T <= 1024;
        if (rs_lazy) return FL_OK;
        return launch_rms_finalize(L, sc.rs_part, re.np, D.eps, sc.inv_rms, T, D.h);
    };
    auto with_parts = [&](Launcher &L, Scratch &sc) {             // the launcher of the projection behind a lazy residual epilogue
        if (rs_lazy) { L.rsp = RsParts{sc.rs_part, gemm_resid_partials(D.h), D.eps, 1.0f / (float)D.h}; rs_lazy = false; }
    };
    for (int64_t l = 0; l < D.L; l++) {
        for (size_t i = 0; i < ns; i++) {
            Shard &sh = m->shards[i]; Scratch &sc = SC(sh); CacheShard &cs = c->shards[i]; LayerW &ly = sh.layers[l];
            FL_HIP(hipSetDevice(sh.device));
            Launcher L = make_launcher(m, sh);
            const int64_t nq = (sh.Hs + 2 * sh.Hkvs) * D.d;
            const size_t kv_layer = (size_t)l * sh.Hkvs * c->seq_alloc * D.d * m->esize();
            void *kc = (char *)cs.k + kv_layer, *vc = (char *)cs.v + kv_layer;
            if (!norm_done) FL_TRY(launch_rmsnorm_add(L, dt, sc.x_res, l == 0 ? nullptr : sc.delta, ly.ln1, D.eps, sc.xn, sc.inv_rms, T, D.h, nslab, slab));
            norm_done = false;
            int qkv_slabs = 1;
            const int qkv_split = tune(TK_QKV_SPLIT);
            // (Qwen2's q/k/v bias moves into the RoPE launch, which sums the slabs anyway: with the bias in the GEMM epilogue the
            // projection could not run in K slices and a mid-size prompt's QKV sat on 128x128 tiles -- T = 512: 63 us at 0.27 PFLOP/s)
            const int64_t sa = (int64_t)c->seq_alloc;
            int h4_qkv = dt == FL_DTYPE_BF16 ? gemm_h4_plan(T, nq, D.h, EPI_QKV_ROPE) : 0;
            if (!h4_qkv && dt == FL_DTYPE_BF16 && m->tp > 1 && D.d * sh.Hkvs % 128 == 0 && nq % 128 == 0) h4_qkv = gemm_h4_plan_whole(T, nq, D.h, EPI_QKV_ROPE);   // (a rank's narrower q | k | v)
            with_parts(L, sc);                                       // (the previous layer's down_proj may have left 1/rms as partial sums)
            const int skf_qkv = dt == FL_DTYPE_BF16 && !h4_qkv && m->tp == 1 && tune(TK_GEMM_SKF) >= 2 ? gemm_skf_plan(T, nq, D.h, EPI_QKV_ROPE, (int)D.d) : 0;
            if (skf_qkv) {
                // short prompts: the same epilogue on the weight-streaming kernel (k_gemm_skf.hip), its K slices met inside the launch
                RopeEpi ro;
                ro.st = cs.st; ro.cos_tab = sh.cos_tab; ro.sin_tab = sh.sin_tab; ro.max_pos = (int)D.max_pos; ro.q_out = sc.q; ro.k_cache = kc; ro.v_cache = vc;
                ro.H = (int)sh.Hs; ro.Hkv = (int)sh.Hkvs; ro.d = (int)D.d; ro.max_seq = (int)sa; ro.v_transposed = c->v_transposed ? 1 : 0;
                FL_TRY(launch_gemm_skf(L, ly.wqkv, sc.xn, ly.bqkv, nullptr, T, nq, D.h, EPI_QKV_ROPE, sc.inv_rms, skf_qkv, nullptr, &ro));
            } else if (h4_qkv) {
                // mid-size prompts: RoPE, bias and the KV append ride in the projection's epilogue (k_gemm_h4.hip): no fp32 QKV matrix
                RopeEpi ro;
                ro.st = cs.st; ro.cos_tab = sh.cos_tab; ro.sin_tab = sh.sin_tab; ro.max_pos = (int)D.max_pos; ro.q_out = sc.q; ro.k_cache = kc; ro.v_cache = vc;
                ro.H = (int)sh.Hs; ro.Hkv = (int)sh.Hkvs; ro.d = (int)D.d; ro.max_seq = (int)sa; ro.v_transposed = c->v_transposed ? 1 : 0;
                FL_TRY(launch_gemm_h4(L, ly.wqkv, sc.xn, ly.bqkv, nullptr, T, nq, D.h, EPI_QKV_ROPE, sc.inv_rms, h4_qkv, nq, nullptr, &ro));
            } else if (dt == FL_DTYPE_BF16 && gemm_qkv_rope_long_plan(T, nq, D.h, std::min(qkv_split, qkv_split_cap(T)))) {
                // long prompts: the same epilogue on the four-wave 256 x 256 kernel (+ the 128 x 256 kernel for peeled tail columns)
                RopeEpi ro;
                ro.st = cs.st; ro.cos_tab = sh.cos_tab; ro.sin_tab = sh.sin_tab; ro.max_pos = (int)D.max_pos; ro.q_out = sc.q; ro.k_cache = kc; ro.v_cache = vc;
                ro.H = (int)sh.Hs; ro.Hkv = (int)sh.Hkvs; ro.d = (int)D.d; ro.max_seq = (int)sa; ro.v_transposed = c->v_transposed ? 1 : 0;
                FL_TRY(launch_gemm_qkv_rope_long(L, ly.wqkv, sc.xn, ly.bqkv, T, nq, D.h, sc.inv_rms, ro, std::min(qkv_split, qkv_split_cap(T))));
            } else {
                FL_TRY(launch_linear(L, dt, ly.wqkv, sc.xn, nullptr, sc.qkv, T, nq, D.h, EPI_F32, sc.inv_rms,
                                     std::min(qkv_split, qkv_split_cap(T)), &qkv_slabs));
                FL_TRY(launch_rope_kv(L, dt, sc.qkv, cs.st, sh.cos_tab, sh.sin_tab, D.max_pos, sc.q, kc, vc, T, sh.Hs, sh.Hkvs, D.d, sa, c->v_transposed, qkv_slabs, ly.bqkv));
            }
            L.rsp = RsParts{};
            if (T == 1) {
                AttnScratch as{cs.part_m, cs.part_l, cs.part_o, cs.counters, c->nsplit, len_hint + 1};
                if (c->v_transposed) FL_TRY(launch_attn_decode_mfma(L, sc.q, kc, vc, cs.st, sc.ao, as, sh.Hs, sh.Hkvs, D.d, sa, D.scale));
                else FL_TRY(launch_attn_decode(L, dt, sc.q, kc, vc, cs.st, sc.ao, as, sh.Hs, sh.Hkvs, D.d, sa, D.scale));
            } else if (c->v_transposed) {
                FL_TRY(launch_attn_prefill_mfma(L, sc.q, kc, vc, cs.st, sc.ao, T, sh.Hs, sh.Hkvs, D.d, sa, D.scale, D.window));
            } else {
                FL_TRY(launch_attn_prefill(L, dt, sc.q, kc, vc, cs.st, sc.ao, T, sh.Hs, sh.Hkvs, D.d, sa, D.scale, D.window));
            }
            if (resid_ok && gemm_resid_supported(dt, T, D.h, sh.Hs * D.d, max_split)) {
                FL_TRY(linear_resid(L, sh, sc, ly.wo, sc.ao, sh.Hs * D.d, ly.ln2, gemm_takes_rs_parts(dt, T, 2 * sh.Ip, D.h, EPI_GATEUP, 1)));
                norm_done = true;
            } else {
                FL_TRY(launch_linear(L, dt, ly.wo, sc.ao, nullptr, sc.delta, T, D.h, sh.Hs * D.d, EPI_F32, nullptr, max_split, &nslab));
            }
        }
        FL_TRY(all_reduce_delta(m, pre, T * D.h));
        for (size_t i = 0; i < ns; i++) {
            Shard &sh = m->shards[i]; Scratch &sc = SC(sh); LayerW &ly = sh.layers[l];
            FL_HIP(hipSetDevice(sh.device));
            Launcher L = make_launcher(m, sh);
            if (!norm_done) FL_TRY(launch_rmsnorm_add(L, dt, sc.x_res, sc.delta, ly.ln2, D.eps, sc.xn, sc.inv_rms, T, D.h, nslab, slab));
            norm_done = false;
            with_parts(L, sc);
            FL_TRY(launch_linear(L, dt, ly.wgu, sc.xn, nullptr, sc.act, T, 2 * sh.Ip, D.h, EPI_GATEUP, sc.inv_rms));
            L.rsp = RsParts{};
            if (resid_ok && gemm_resid_supported(dt, T, D.h, sh.Ip, max_split)) {
                // (the next layer's QKV projection takes the partial sums if its kernel can; the last layer's final norm wants the vector)
                const int64_t nq = (sh.Hs + 2 * sh.Hkvs) * D.d;
                const bool next_takes = l + 1 < D.L && (gemm_h4_plan(T, nq, D.h, EPI_QKV_ROPE) > 0 || (tune(TK_GEMM_SKF) >= 2 && gemm_skf_plan(T, nq, D.h, EPI_QKV_ROPE, (int)D.d) > 0) ||
                                                        gemm_takes_rs_parts(dt, T, nq, D.h, EPI_F32, std::min(tune(TK_QKV_SPLIT), qkv_split_cap(T))));
                FL_TRY(linear_resid(L, sh, sc, ly.wd, sc.act, sh.Ip, l + 1 < D.L ? sh.layers[l + 1].ln1 : sh.norm, next_takes));
                norm_done = true;
            } else {
                FL_TRY(launch_linear(L, dt, ly.wd, sc.act, nullptr, sc.delta, T, D.h, sh.Ip, EPI_F32, nullptr, max_split, &nslab));
            }
        }
        FL_TRY(all_reduce_delta(m, pre, T * D.h));
    }
    // narrow(1, T-1, 1) -> final norm -> lm_head on the last position only (K12)
    for (size_t i = 0; i < ns; i++) {
        Shard &sh = m->shards[i]; Scratch &sc = SC(sh);
        FL_HIP(hipSetDevice(sh.device));
        Launcher L = make_launcher(m, sh);
        float *xl = sc.x_res + (size_t)(T - 1) * D.h, *dl = sc.delta + (size_t)(T - 1) * D.h;
        void *xnl = (char *)sc.xn + (size_t)(T - 1) * D.h * m->esize();
        if (!norm_done) FL_TRY(launch_rmsnorm_add(L, dt, xl, dl, sh.norm, D.eps, xnl, sc.inv_rms + (T - 1), 1, D.h, nslab, slab));
        FL_TRY(launch_linear(L, dt, sh.lm_head, xnl, nullptr, lm_head_out(m, sh), 1, sh.Vs, D.h, EPI_F32, sc.inv_rms + (T - 1)));
    }
    FL_TRY(gather_logits(m));
    return FL_OK;
}

// sampler: non-null (re)sets the cache's token selection; null keeps it (later chunks of one call)
static int set_state(Model *m, Cache *c, uint32_t token, size_t pos, size_t len, uint32_t step, int64_t eos, size_t call0 = (size_t)-1,
                     const SampleState *sampler = nullptr) {
    if (call0 == (size_t)-1) call0 = len;
    for (size_t i = 0; i < m->shards.size(); i++) {
        Shard &sh = m->shards[i];
        FL_HIP(hipSetDevice(sh.device));
        hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(64), 0, sh.stream, c->shards[i].st, token, (uint32_t)pos,
                           (uint32_t)len, (uint32_t)call0, step, (int32_t)eos, c->shards[i].heads_done, (int)(m->D.L * sh.Hkvs),
                           c->shards[i].ss, sampler ? *sampler : SampleState{}, sampler ? 1 : 0);
        FL_HIP(hipGetLastError());
    }
    return FL_OK;
}

static int enqueue_argmax(Model *m, Cache *c, int advance) {
    for (size_t i = 0; i < m->shards.size(); i++) {
        Shard &sh = m->shards[i];
        FL_HIP(hipSetDevice(sh.device));
        Launcher L = make_launcher(m, sh);
        FL_TRY(launch_select_advance(L, sh.logits_full, m->D.V, c->shards[i].st, c->shards[i].ss, c->shards[i].sel_scratch, c->shards[i].out_tokens, advance,
                                     sh.amax_valid ? sh.amax : nullptr, sh.eng_epoch));
    }
    return FL_OK;
}

static int sync_all(Model *m) {
    for (auto &sh : m->shards) { FL_HIP(hipSetDevice(sh.device)); FL_HIP(hipStreamSynchronize(sh.stream)); }
    return FL_OK;
}

// One decode step (embed .. lm_head .. argmax+advance) reading everything from the device state.
// Single-shard models replay it as a hipGraph (the ~11 launches per layer are launch-bound at
// TinyLlama scale); captured lazily on the second step of a cache so that all lazy module /
// attribute initialisation has already happened eagerly.
static int decode_step(Model *m, Cache *c, int64_t len_hint) {
    // Graphs: one shard per process (plain single GPU, or one rank of a multi-process TP group: RCCL collectives are
    // stream-ordered and capturable, every rank captures the same sequence); or all shards of a single-process group
    // whose collectives are one-shot (they synchronise through memory, so each shard's step is its own graph).
    const int tp_graph = tune(TK_TP_GRAPH);
    const size_t ns = m->shards.size();
    const bool one_shard = ns == 1 && (m->tp == 1 || (m->tp_mode == FL_TP_MULTI_PROCESS && tp_graph));
    const bool local_group = ns > 1 && m->tp_mode == FL_TP_SINGLE_PROCESS && tp_graph && m->shards[0].pc.connected &&
                             m->D.h <= m->shards[0].pc.nmax && m->shards[0].Vs <= m->shards[0].pc.nmax;
    const bool graphable = m->use_graph && !m->profiling && (one_shard || local_group) && !c->graph_failed;
    if (graphable && c->shards[0].graph) {
        for (size_t i = 0; i < ns; i++) {
            FL_HIP(hipSetDevice(m->shards[i].device));
            FL_HIP(hipGraphLaunch(c->shards[i].graph, m->shards[i].stream));
        }
        return FL_OK;
    }
    if (graphable && c->warm_steps >= 1) {
        std::vector<hipGraph_t> gs(ns, nullptr);
        bool ok = true;
        size_t begun = 0;
        for (; begun < ns && ok; begun++) {
            FL_HIP(hipSetDevice(m->shards[begun].device));
            ok = hipStreamBeginCapture(m->shards[begun].stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (!ok) break;
        }
        int rc = FL_OK;
        if (ok) {
            rc = enqueue_forward(m, c, false, 1, false, len_hint);
            if (rc == FL_OK) rc = enqueue_argmax(m, c, 1);
        }
        for (size_t i = 0; i < begun; i++) {
            (void)hipSetDevice(m->shards[i].device);
            const hipError_t e = hipStreamEndCapture(m->shards[i].stream, &gs[i]);
            ok = ok && rc == FL_OK && e == hipSuccess && gs[i] != nullptr;
        }
        for (size_t i = 0; i < ns && ok; i++) {
            (void)hipSetDevice(m->shards[i].device);
            ok = hipGraphInstantiate(&c->shards[i].graph, gs[i], nullptr, nullptr, 0) == hipSuccess;
        }
        for (auto g : gs) if (g) (void)hipGraphDestroy(g);
        if (ok) {
            for (size_t i = 0; i < ns; i++) {
                FL_HIP(hipSetDevice(m->shards[i].device));
                FL_HIP(hipGraphLaunch(c->shards[i].graph, m->shards[i].stream));
            }
            return FL_OK;
        }
        (void)hipGetLastError();
        for (size_t i = 0; i < ns; i++) if (c->shards[i].graph) { (void)hipGraphExecDestroy(c->shards[i].graph); c->shards[i].graph = nullptr; }
        c->graph_failed = true;                                     // fall through to eager launches
    }
    FL_TRY(enqueue_forward(m, c, false, 1, false, len_hint));
    FL_TRY(enqueue_argmax(m, c, 1));
    c->warm_steps++;
    return FL_OK;
}

static int check_call(Model *m, Cache *c, size_t T, size_t pos) {
    if (!m || !c) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null model or cache");
    if (c->m != m) FL_FAIL(FL_ERR_BAD_ARGUMENT, "cache belongs to another model");
    if (T == 0) FL_FAIL(FL_ERR_BAD_ARGUMENT, "empty input");
    if (c->len + T > c->max_seq) FL_FAIL(FL_ERR_SEQ_OVERFLOW, "sequence overflow: %zu cached + %zu new > capacity %zu", c->len, T, c->max_seq);
    if (pos + T > (size_t)m->D.max_pos) FL_FAIL(FL_ERR_SEQ_OVERFLOW, "position %zu exceeds max_position_embeddings %lld", pos + T, (long long)m->D.max_pos);
    return FL_OK;
}

int forward(Model *m, Cache *c, const uint32_t *ids, size_t T, size_t pos, float *logits_out, uint32_t *token_out,
            const fl_sampling *sampling) {
    FL_TRY(check_call(m, c, T, pos));
    debug_inject("forward");
    const SampleState sampler = make_sampler(sampling);
    if (!ids) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null ids");
    for (size_t t = 0; t < T; t++)
        if ((int64_t)ids[t] >= m->D.V) FL_FAIL(FL_ERR_BAD_ARGUMENT, "token id %u out of range (vocab %lld)", ids[t], (long long)m->D.V);
    std::lock_guard<std::mutex> lock(m->mu);
    const Dims &D = m->D;
    if (T == 1) {
        FL_TRY(set_state(m, c, ids[0], pos, c->len, 0, -1, (size_t)-1, &sampler));
        FL_TRY(decode_step(m, c, (int64_t)c->len));
        c->len += 1;
    } else {
        const int64_t chunk_max = tune(TK_PREFILL_CHUNK);
        size_t done = 0;
        const size_t call0 = c->len;                 // the mask of every chunk is that of the single call
        while (done < T) {
            int64_t Tc = (int64_t)std::min<size_t>(T - done, (size_t)chunk_max);
            if (T - done - (size_t)Tc == 1 && Tc > 2) Tc -= 1;      // never leave a 1-token tail (it would take the decode mask)
            for (auto &sh : m->shards) {
                FL_HIP(hipSetDevice(sh.device));
                if (sh.pre.cap_T < Tc) FL_TRY(grow_prefill_scratch(m, sh, Tc));
                FL_HIP(hipMemcpyAsync(sh.pre.ids, ids + done, (size_t)Tc * 4, hipMemcpyHostToDevice, sh.stream));
            }
            FL_TRY(set_state(m, c, ids[done], pos + done, c->len, 0, -1, call0, &sampler));
            if (Tc == 1) {
                // a 1-token tail chunk goes through the decode kernels but is still one `forward`
                FL_TRY(enqueue_forward(m, c, false, 1, false, (int64_t)c->len));
            } else {
                FL_TRY(enqueue_forward(m, c, true, Tc, true, (int64_t)c->len));
            }
            c->len += (size_t)Tc;
            done += (size_t)Tc;
        }
        FL_TRY(enqueue_argmax(m, c, 0));
    }
    Shard &s0 = m->shards[0];
    FL_HIP(hipSetDevice(s0.device));
    if (logits_out) FL_HIP(hipMemcpyAsync(m->host_logits, s0.logits_full, (size_t)D.V * 4, hipMemcpyDeviceToHost, s0.stream));
    if (token_out) FL_HIP(hipMemcpyAsync(m->host_tokens, c->shards[0].out_tokens, 4, hipMemcpyDeviceToHost, s0.stream));
    FL_HIP(hipMemcpyAsync(m->host_state, c->shards[0].st, sizeof(StepState), hipMemcpyDeviceToHost, s0.stream));
    FL_TRY(sync_all(m));
    if (m->host_state->error) FL_FAIL(FL_ERR_HIP, "device-side wait gave up (code 0x%x): decode kernels did not make progress", m->host_state->error);
    FL_TRY(comm_check(m));
    if (logits_out) memcpy(logits_out, m->host_logits, (size_t)D.V * 4);
    if (token_out) *token_out = m->host_tokens[0];
    return FL_OK;
}

int decode_greedy(Model *m, Cache *c, uint32_t first, size_t pos, size_t n_steps, int64_t eos,
                  uint32_t *tokens_out, size_t *n_out, const fl_sampling *sampling) {
    const SampleState sampler = make_sampler(sampling);
    if (n_out) *n_out = 0;
    if (n_steps == 0) return FL_OK;
    FL_TRY(check_call(m, c, n_steps, pos));
    if (!tokens_out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null tokens_out");
    if ((int64_t)first >= m->D.V) FL_FAIL(FL_ERR_BAD_ARGUMENT, "token id %u out of range", first);
    std::lock_guard<std::mutex> lock(m->mu);
    Shard &s0 = m->shards[0];
    size_t done = 0;
    uint32_t tok = first;
    // With an EOS id the host looks at the tokens between chunks of 16, 32, ... 256 steps: the reference's loop breaks at
    // the first EOS (mod.rs:431-436), and a request that ends after 20 tokens must not pay for n_steps forwards.  Without
    // one, a chunk is as long as the token buffer allows (one sync per 4096 steps).
    size_t chunk = eos >= 0 ? 16 : kOutTokensCap;
    while (done < n_steps) {
        const size_t nb = std::min(n_steps - done, chunk);
        if (eos >= 0) chunk = std::min<size_t>(chunk * 2, 256);
        FL_TRY(set_state(m, c, tok, pos + done, c->len, 0, eos, (size_t)-1, done == 0 ? &sampler : nullptr));
        for (size_t i = 0; i < nb; i++) FL_TRY(decode_step(m, c, (int64_t)(c->len + i)));
        FL_HIP(hipSetDevice(s0.device));
        FL_HIP(hipMemcpyAsync(m->host_tokens, c->shards[0].out_tokens, nb * 4, hipMemcpyDeviceToHost, s0.stream));
        FL_HIP(hipMemcpyAsync(m->host_state, c->shards[0].st, sizeof(StepState), hipMemcpyDeviceToHost, s0.stream));
        FL_TRY(sync_all(m));
        if (m->host_state->error) FL_FAIL(FL_ERR_HIP, "device-side wait gave up (code 0x%x): decode kernels did not make progress", m->host_state->error);
        FL_TRY(comm_check(m));
        for (size_t i = 0; i < nb; i++) {
            const uint32_t t = m->host_tokens[i];
            if (eos >= 0 && (int64_t)t == eos) {
                // the forward that produced EOS was needed; what ran after it is discarded
                c->len += i + 1;
                if (n_out) *n_out = done + i;
                return FL_OK;
            }
            tokens_out[done + i] = t;
        }
        c->len += nb;
        tok = m->host_tokens[nb - 1];
        done += nb;
    }
    if (n_out) *n_out = done;
    return FL_OK;
}

// ------------------------------------------------------------------------------- batched decode (row N4)
Batch::~Batch() {
    if (!m) return;
    std::lock_guard<std::mutex> lock(m->mu);
    Shard &sh = m->shards[0];
    (void)hipSetDevice(sh.device);
    (void)hipStreamSynchronize(sh.stream);
    if (graph) (void)hipGraphExecDestroy(graph);
    for (void *p : allocs) (void)hipFree(p);
    if (host_tokens) (void)hipHostFree(host_tokens);
    if (host_states) (void)hipHostFree(host_states);
}

int batch_create(Model *m, Cache *const *caches, size_t B, Batch **out) {
    if (!m || !caches || !out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
    if (B < 1 || B > (size_t)kMaxBatch) FL_FAIL(FL_ERR_BAD_ARGUMENT, "batch size %zu not in 1..%d", B, kMaxBatch);
    // one GPU, or -- round 5 -- one rank of a multi-process tensor-parallel group (every rank builds the same batch and calls the same
    // entry points: the step's all-reduces and logits gather are collectives)
    const bool tp_rank = m->tp > 1 && m->tp_mode == FL_TP_MULTI_PROCESS && m->shards.size() == 1;
    if (m->shards.size() != 1 || (m->tp != 1 && !tp_rank)) FL_FAIL(FL_ERR_UNSUPPORTED, "batched decode runs on one GPU or on the ranks of an FL_TP_MULTI_PROCESS group");
    if (tp_rank && (!m->shards[0].pc.connected || !m->vocab_parallel || m->shards[0].Vs % 4))
        FL_FAIL(FL_ERR_UNSUPPORTED, "batched decode on a tensor-parallel group needs connected peer inboxes and a vocabulary shard that is a multiple of 4");
    const Dims &D = m->D;
    Shard &sh = m->shards[0];
    // fp32 models (the literal-parity mode) and caches without the MFMA attention layout: the weights are still read once per step for
    // all B rows; embedding, RoPE / KV append and attention run as the single-sequence kernels on row i of the batch (3 B small launches
    // per layer, replayed from the step's graph)
    bool per_seq = m->dtype != FL_DTYPE_BF16;
    for (size_t i = 0; i < B; i++) {
        if (!caches[i] || caches[i]->m != m) FL_FAIL(FL_ERR_BAD_ARGUMENT, "cache %zu is null or belongs to another model", i);
        if (!caches[i]->v_transposed) per_seq = true;
        for (size_t j = 0; j < i; j++) if (caches[j] == caches[i]) FL_FAIL(FL_ERR_BAD_ARGUMENT, "cache %zu appears twice in the batch", i);
    }
    if (per_seq && tp_rank) FL_FAIL(FL_ERR_UNSUPPORTED, "batched decode on a tensor-parallel group is bf16 with the MFMA attention layout (head_dim 64/128, group <= 8)");
    std::unique_ptr<Batch> b(new Batch());
    std::lock_guard<std::mutex> lock(m->mu);
    b->m = m; b->B = (int)B; b->per_seq = per_seq;
    if (per_seq && attn_decode_batch_supported(D.d)) {
        b->plain = true;
        for (size_t i = 0; i < B; i++) if (caches[i]->v_transposed) b->plain = false;
    }
    b->caches.assign(caches, caches + B);
    // B >= 3: the prefill-shaped step (separate norm / RoPE launches) with the wide projections on the LDS-DMA ring kernel
    b->dma = !per_seq && B >= (size_t)tune(TK_BATCH_DMA_MIN) && gemv_dma_supported((int)B, 2 * sh.Ip, D.h, EPI_GATEUP, 0) &&
             gemv_dma_supported((int)B, sh.Vs, D.h, EPI_F32, 0) && gemv_dma_ksplit(D.h, 0, EPI_GATEUP) == 1;
    const bool gemv_rows = B <= (size_t)kMaxBatchGemv && !tp_rank && !per_seq;   // the streaming GEMV forms hold at most eight rows (and know no all-reduce)
    if (gemv_rows) {
        b->nks_o = gemv_batch_ksplit((int)B, sh.Hs * D.d, D.h, EPI_F32);
        b->nks_down = gemv_batch_ksplit((int)B, sh.Ip, D.h, EPI_F32);
        if (gemv_batch_ksplit((int)B, D.h, 2 * sh.Ip, EPI_GATEUP) != 1) FL_FAIL(FL_ERR_UNSUPPORTED, "hidden size %lld too large for the batched norm prologue", (long long)D.h);
    }
    FL_HIP(hipSetDevice(sh.device));
    std::vector<SeqRef> refs(B);
    for (size_t i = 0; i < B; i++) {
        CacheShard &cs = caches[i]->shards[0];
        refs[i] = SeqRef{cs.st, cs.ss, cs.k, cs.v, cs.part_m, cs.part_l, cs.part_o, cs.counters, cs.out_tokens, cs.sel_scratch,
                         (int)caches[i]->seq_alloc, caches[i]->nsplit};
        b->max_nsplit = std::max(b->max_nsplit, caches[i]->nsplit);
    }
    const size_t es = m->esize();
    const int nsl = std::max(b->nks_o, b->nks_down);
    FL_TRY(dev_alloc(b->allocs, (void **)&b->seqs_dev, sizeof(SeqRef) * B, nullptr));
    FL_TRY(dev_alloc(b->allocs, (void **)&b->x_res, B * D.h * 4, nullptr));
    FL_TRY(dev_alloc(b->allocs, (void **)&b->x_res2, B * D.h * 4, nullptr));
    FL_TRY(dev_alloc(b->allocs, (void **)&b->delta, (size_t)nsl * B * D.h * 4, nullptr));
    FL_TRY(dev_alloc(b->allocs, &b->q, B * sh.Hs * D.d * es, nullptr));
    FL_TRY(dev_alloc(b->allocs, &b->ao, B * sh.Hs * D.d * es, nullptr));
    FL_TRY(dev_alloc(b->allocs, &b->act, B * sh.Ip * es, nullptr));
    FL_TRY(dev_alloc(b->allocs, (void **)&b->logits, B * D.V * 4, nullptr));
    if (tp_rank) {
        FL_TRY(dev_alloc(b->allocs, (void **)&b->logits_local, B * sh.Vs * 4, nullptr));
        FL_TRY(dev_alloc(b->allocs, (void **)&b->logits_ranks, B * D.V * 4, nullptr));
    }
    // B >= 7: every projection through the short-prompt GEMM with the norm and RoPE / KV append as their own small
    // launches, i.e. the prefill pipeline at T = B with per-sequence positions and caches.  Measured (Mistral-7B,
    // ms per step, unfused vs fused): B = 3 4.16 / 3.87, 4 4.21 / 3.99, 6 4.24 / 4.18, 8 4.26 / 4.38
    b->unfused = B >= (size_t)(tune(TK_BATCH_UNFUSED_MIN) >= 0 ? tune(TK_BATCH_UNFUSED_MIN) : (b->dma ? 3 : 7)) && gemm_skinny_supported((int64_t)B, D.h, D.h) &&
                 gemm_skinny_supported((int64_t)B, D.h, sh.Ip);
    // more than eight streams, or a tensor-parallel rank: always the prefill-shaped step (launch_linear finds a kernel for every shape)
    if (!gemv_rows) b->unfused = true;
    if (b->unfused) {
        FL_TRY(alloc_scratch(m, sh, b->sc, (int64_t)B, &b->allocs));
    }
    FL_HIP(hipMemcpyAsync(b->seqs_dev, refs.data(), sizeof(SeqRef) * B, hipMemcpyHostToDevice, sh.stream));
    FL_HIP(hipStreamSynchronize(sh.stream));                  // refs is a stack vector
    FL_HIP(hipHostMalloc((void **)&b->host_tokens, B * kBatchChunk * 4, hipHostMallocDefault));
    FL_HIP(hipHostMalloc((void **)&b->host_states, B * sizeof(StepState), hipHostMallocDefault));
    m->refs.fetch_add(1);
    *out = b.release();
    return FL_OK;
}

// Continuous batching: sequence `slot` of a batch leaves (EOS, cancelled) and another stream's cache takes its place,
// without rebuilding the batch.  Every kernel of the step reads a sequence's pointers, length and split count from its SeqRef in
// device memory, so the step's captured graph stays valid: the swap is one 80-byte copy.  The graph is dropped (and re-captured by
// the next step) only where launch geometry or node arguments depend on the caches: a larger attention split count than any
// sequence had so far, or the per-sequence launches of a mixed-layout batch.
int batch_replace(Batch *b, size_t slot, Cache *c) {
    if (!b || !c) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
    Model *m = b->m;
    if (slot >= (size_t)b->B) FL_FAIL(FL_ERR_BAD_ARGUMENT, "slot %zu not in 0..%d", slot, b->B - 1);
    if (c->m != m) FL_FAIL(FL_ERR_BAD_ARGUMENT, "the cache belongs to another model");
    for (int i = 0; i < b->B; i++)
        if (b->caches[i] == c && (size_t)i != slot) FL_FAIL(FL_ERR_BAD_ARGUMENT, "the cache is sequence %d of this batch already", i);
    if (b->caches[slot] == c) return FL_OK;
    if (!b->per_seq && !c->v_transposed) FL_FAIL(FL_ERR_UNSUPPORTED, "this batch runs the MFMA batch attention: the new cache must be in that layout too");
    if (b->plain && c->v_transposed) FL_FAIL(FL_ERR_UNSUPPORTED, "this batch runs the plain-layout batch attention: the new cache must be in that layout too");
    std::lock_guard<std::mutex> lock(m->mu);
    Shard &sh = m->shards[0];
    FL_HIP(hipSetDevice(sh.device));
    FL_HIP(hipStreamSynchronize(sh.stream));
    CacheShard &cs = c->shards[0];
    const SeqRef ref{cs.st, cs.ss, cs.k, cs.v, cs.part_m, cs.part_l, cs.part_o, cs.counters, cs.out_tokens, cs.sel_scratch, (int)c->seq_alloc, c->nsplit};
    FL_HIP(hipMemcpy(b->seqs_dev + slot, &ref, sizeof(SeqRef), hipMemcpyHostToDevice));
    b->caches[slot] = c;
    const bool regraph = c->nsplit > b->max_nsplit || (b->per_seq && !b->plain);
    b->max_nsplit = std::max(b->max_nsplit, c->nsplit);
    if (regraph && b->graph) { (void)hipGraphExecDestroy(b->graph); b->graph = nullptr; }
    return FL_OK;
}

static int enqueue_batch_step_unfused(Batch *b);

// One decode step of the whole batch: the 5-launch layer of enqueue_decode_fused with B activation rows.
static int enqueue_batch_step(Batch *b) {
    if (b->unfused) return enqueue_batch_step_unfused(b);
    Model *m = b->m;
    const Dims &D = m->D;
    Shard &sh = m->shards[0];
    Launcher L = make_launcher(m, sh);
    const int B = b->B;
    const long long slab = (long long)B * D.h;
    for (int64_t l = 0; l < D.L; l++) {
        LayerW &ly = sh.layers[l];
        GemvBatchArgs a;
        a.B = B; a.seqs = b->seqs_dev;
        a.W = ly.wqkv; a.bias = ly.bqkv; a.N = (int)((sh.Hs + 2 * sh.Hkvs) * D.d); a.K = (int)D.h; a.nks = 1;
        a.epi = EPI_QKV_ROPE; a.pro = PRO_NORM; a.norm_w = ly.ln1; a.eps = D.eps;
        if (l == 0) { a.embed = sh.embed; a.x_out = b->x_res2; }
        else { a.x_in = b->x_res; a.delta = b->delta; a.n_slab = b->nks_down; a.slab_stride = slab; a.x_out = b->x_res2; }
        a.cos_tab = sh.cos_tab; a.sin_tab = sh.sin_tab; a.q_out = b->q; a.kv_layer_off = (size_t)l * sh.Hkvs * D.d;
        a.H = (int)sh.Hs; a.Hkv = (int)sh.Hkvs; a.d = (int)D.d; a.max_pos = (int)D.max_pos;
        FL_TRY(launch_gemv_batch(L, a));
        FL_TRY(launch_attn_decode_mfma_batch(L, b->q, b->seqs_dev, B, b->max_nsplit, (size_t)l * sh.Hkvs * D.d, b->ao, sh.Hs, sh.Hkvs,
                                             D.d, D.scale, 0.0));
        GemvBatchArgs o;
        o.B = B; o.W = ly.wo; o.x = b->ao; o.out = b->delta; o.N = (int)D.h; o.K = (int)(sh.Hs * D.d); o.nks = b->nks_o;
        FL_TRY(launch_gemv_batch(L, o));
        GemvBatchArgs g;
        g.B = B; g.seqs = b->seqs_dev; g.W = ly.wgu; g.out = b->act; g.N = (int)(2 * sh.Ip); g.K = (int)D.h; g.epi = EPI_GATEUP; g.pro = PRO_NORM;
        g.x_in = b->x_res2; g.delta = b->delta; g.n_slab = b->nks_o; g.slab_stride = slab; g.norm_w = ly.ln2; g.eps = D.eps; g.x_out = b->x_res;
        FL_TRY(launch_gemv_batch(L, g));
        GemvBatchArgs d;
        d.B = B; d.W = ly.wd; d.x = b->act; d.out = b->delta; d.N = (int)D.h; d.K = (int)sh.Ip; d.nks = b->nks_down;
        FL_TRY(launch_gemv_batch(L, d));
    }
    GemvBatchArgs h;
    h.B = B; h.seqs = b->seqs_dev; h.W = sh.lm_head; h.out = b->logits; h.N = (int)D.V; h.K = (int)D.h; h.pro = PRO_NORM;
    h.x_in = b->x_res; h.delta = b->delta; h.n_slab = b->nks_down; h.slab_stride = slab; h.norm_w = sh.norm; h.eps = D.eps;
    FL_TRY(launch_gemv_batch(L, h));
    return launch_select_advance_batch(L, b->logits, D.V, b->seqs_dev, B, 1);
}

// The same step as 8 launches per layer: rmsnorm_add -> QKV GEMM -> RoPE / KV append -> attention -> o_proj GEMM (K
// slabs) -> rmsnorm_add (sums them) -> gate/up GEMM -> down GEMM (K slabs); launch_linear picks gemm_skinny_kernel.
static int enqueue_batch_step_unfused(Batch *b) {
    Model *m = b->m;
    const Dims &D = m->D;
    Shard &sh = m->shards[0];
    Scratch &sc = b->sc;
    Launcher L = make_launcher(m, sh);
    const int dt = m->dtype, B = b->B;
    const int64_t T = B, slab = T * D.h;
    int nslab = 1;
    // the two wide projections (gate/up, lm_head: thousands of 16-row units) stream fastest through the LDS-DMA ring kernel;
    // the narrow ones (QKV, o_proj, down_proj: one or two units per CU) through K slices of the short-prompt GEMM
    auto wide = [&](const void *W, void *out, int64_t N, int epi) -> int {
        if (!b->dma) return launch_linear(L, dt, W, sc.xn, nullptr, out, T, N, D.h, epi, sc.inv_rms);
        GemvBatchArgs ga;
        ga.W = W; ga.x = sc.xn; ga.x_scale = sc.inv_rms; ga.out = out; ga.N = (int)N; ga.K = (int)D.h; ga.epi = epi; ga.pro = PRO_X; ga.B = B; ga.nks = 1;
        return launch_gemv_dma(L, ga);
    };
    const bool ps = b->per_seq;
    const size_t es = m->esize();
    if (ps && !b->plain) { for (int i = 0; i < B; i++) FL_TRY(launch_embed(L, dt, sh.embed, nullptr, b->caches[i]->shards[0].st, sc.x_res + (size_t)i * D.h, 1, D.h)); }
    else FL_TRY(launch_embed_batch(L, sh.embed, b->seqs_dev, sc.x_res, B, D.h, dt));
    const int64_t nq = (sh.Hs + 2 * sh.Hkvs) * D.d;
    // per-sequence mode: row i's RoPE / KV append and attention on the single-sequence kernels (position, length and cache of sequence i)
    auto rope_attn_per_seq = [&](int64_t l, const float *bias) -> int {
        for (int i = 0; i < B; i++) {
            Cache *ci = b->caches[i];
            CacheShard &cs = ci->shards[0];
            const int64_t sa = ci->seq_alloc;
            const size_t kv_layer = (size_t)l * sh.Hkvs * sa * D.d * es;
            void *kc = (char *)cs.k + kv_layer, *vc = (char *)cs.v + kv_layer;
            void *qi = (char *)sc.q + (size_t)i * sh.Hs * D.d * es, *aoi = (char *)sc.ao + (size_t)i * sh.Hs * D.d * es;
            FL_TRY(launch_rope_kv(L, dt, sc.qkv + (size_t)i * nq, cs.st, sh.cos_tab, sh.sin_tab, D.max_pos, qi, kc, vc, 1, sh.Hs, sh.Hkvs, D.d, sa, ci->v_transposed, 1, bias));
            AttnScratch as{cs.part_m, cs.part_l, cs.part_o, cs.counters, ci->nsplit, 0};
            if (ci->v_transposed) FL_TRY(launch_attn_decode_mfma(L, qi, kc, vc, cs.st, aoi, as, sh.Hs, sh.Hkvs, D.d, sa, D.scale));
            else FL_TRY(launch_attn_decode(L, dt, qi, kc, vc, cs.st, aoi, as, sh.Hs, sh.Hkvs, D.d, sa, D.scale));
        }
        return FL_OK;
    };
    // Round 5, FL_GEMM_SKF=2 (off by default: it measured 8-13 % SLOWER, profiles/r05/README.md): the layer as FIVE launches (k_gemm_skf.hip) -- QKV with each row's RoPE / KV append in its epilogue, attention, o_proj and
    // down_proj with the residual + next norm in theirs (K slices met inside the launch: no slabs, no rmsnorm_add), gate/up with its
    // row scales from the partial sums -- where every projection of the model has a plan there; otherwise the eight-launch layer below
    const bool tpr = m->tp > 1;                                        // a rank of a multi-process group: all-reduce behind o_proj / down_proj, gathered logits
    const int ks_q = dt == FL_DTYPE_BF16 && sc.rs_part && !tpr && !ps && tune(TK_GEMM_SKF) >= 2 ? gemm_skf_plan(T, nq, D.h, EPI_QKV_ROPE, (int)D.d) : 0;
    const int ks_o = ks_q ? gemm_skf_plan(T, D.h, sh.Hs * D.d, EPI_RESID) : 0, ks_d = ks_o ? gemm_skf_plan(T, D.h, sh.Ip, EPI_RESID) : 0;
    if (ks_q && ks_o && ks_d && gemm_skf_plan(T, 2 * sh.Ip, D.h, EPI_GATEUP) > 0 && tune(TK_GEMM_RESID)) {
        const int np = gemm_resid_partials(D.h);
        auto resid = [&](const void *W, const void *x, int64_t K, const float *next_w, int ks) -> int {
            ResidEpi re;
            re.h = sc.x_res; re.w = next_w; re.xn = sc.xn; re.part = sc.rs_part; re.np = np;
            return launch_gemm_skf(L, W, x, nullptr, nullptr, T, D.h, K, EPI_RESID, nullptr, ks, &re);
        };
        const RsParts parts{sc.rs_part, np, D.eps, 1.0f / (float)D.h};
        FL_TRY(launch_rmsnorm_add(L, dt, sc.x_res, nullptr, sh.layers[0].ln1, D.eps, sc.xn, sc.inv_rms, T, D.h, 1, slab));
        for (int64_t l = 0; l < D.L; l++) {
            LayerW &ly = sh.layers[l];
            RopeEpi ro;
            ro.cos_tab = sh.cos_tab; ro.sin_tab = sh.sin_tab; ro.max_pos = (int)D.max_pos; ro.q_out = sc.q;
            ro.H = (int)sh.Hs; ro.Hkv = (int)sh.Hkvs; ro.d = (int)D.d; ro.v_transposed = 1;
            ro.seqs = b->seqs_dev; ro.kv_layer_off = (size_t)l * sh.Hkvs * D.d;
            if (l > 0) L.rsp = parts;                                    // (the previous layer's down_proj left 1/rms as partial sums)
            FL_TRY(launch_gemm_skf(L, ly.wqkv, sc.xn, ly.bqkv, nullptr, T, nq, D.h, EPI_QKV_ROPE, sc.inv_rms, ks_q, nullptr, &ro));
            L.rsp = RsParts{};
            FL_TRY(launch_attn_decode_mfma_batch(L, sc.q, b->seqs_dev, B, b->max_nsplit, (size_t)l * sh.Hkvs * D.d, sc.ao, sh.Hs, sh.Hkvs,
                                                 D.d, D.scale, 0.0));
            FL_TRY(resid(ly.wo, sc.ao, sh.Hs * D.d, ly.ln2, ks_o));
            if (b->dma) {                                                // (eight rows at most: the LDS-DMA ring kernel streams gate/up fastest, and takes a vector)
                FL_TRY(launch_rms_finalize(L, sc.rs_part, np, D.eps, sc.inv_rms, T, D.h));
                FL_TRY(wide(ly.wgu, sc.act, 2 * sh.Ip, EPI_GATEUP));
            } else {
                L.rsp = parts;
                FL_TRY(launch_linear(L, dt, ly.wgu, sc.xn, nullptr, sc.act, T, 2 * sh.Ip, D.h, EPI_GATEUP, sc.inv_rms));
                L.rsp = RsParts{};
            }
            FL_TRY(resid(ly.wd, sc.act, sh.Ip, l + 1 < D.L ? sh.layers[l + 1].ln1 : sh.norm, ks_d));
        }
        FL_TRY(launch_rms_finalize(L, sc.rs_part, np, D.eps, sc.inv_rms, T, D.h));
        FL_TRY(wide(sh.lm_head, b->logits, D.V, EPI_F32));
        return launch_select_advance_batch(L, b->logits, D.V, b->seqs_dev, B, 1);
    }
    for (int64_t l = 0; l < D.L; l++) {
        LayerW &ly = sh.layers[l];
        FL_TRY(launch_rmsnorm_add(L, dt, sc.x_res, l == 0 ? nullptr : sc.delta, ly.ln1, D.eps, sc.xn, sc.inv_rms, T, D.h, nslab, slab));
        // K slices for the QKV stream too (96 strips of 64 rows otherwise: a third of the chip); the bias, if any, moves
        // into the RoPE launch, which sums the slabs anyway
        int qkv_slabs = 1;
        if (b->plain) {
            // plain cache layout (fp32 models; bf16 outside the MFMA attention's head shapes): the batch kernels' plain forms
            FL_TRY(launch_linear(L, dt, ly.wqkv, sc.xn, nullptr, sc.qkv, T, nq, D.h, EPI_F32, sc.inv_rms));
            FL_TRY(launch_rope_kv_batch(L, sc.qkv, b->seqs_dev, sh.cos_tab, sh.sin_tab, D.max_pos, sc.q, (size_t)l * sh.Hkvs * D.d, B, sh.Hs,
                                        sh.Hkvs, D.d, 1, ly.bqkv, dt, false));
            FL_TRY(launch_attn_decode_batch(L, dt, sc.q, b->seqs_dev, B, b->max_nsplit, (size_t)l * sh.Hkvs * D.d, sc.ao, sh.Hs, sh.Hkvs, D.d, D.scale));
        } else if (ps) {
            FL_TRY(launch_linear(L, dt, ly.wqkv, sc.xn, nullptr, sc.qkv, T, nq, D.h, EPI_F32, sc.inv_rms));
            FL_TRY(rope_attn_per_seq(l, ly.bqkv));
        } else {
            FL_TRY(launch_linear(L, dt, ly.wqkv, sc.xn, nullptr, sc.qkv, T, nq, D.h, EPI_F32, sc.inv_rms, kMaxQkvSplitShort, &qkv_slabs));
            FL_TRY(launch_rope_kv_batch(L, sc.qkv, b->seqs_dev, sh.cos_tab, sh.sin_tab, D.max_pos, sc.q, (size_t)l * sh.Hkvs * D.d, B, sh.Hs,
                                        sh.Hkvs, D.d, qkv_slabs, ly.bqkv));
            FL_TRY(launch_attn_decode_mfma_batch(L, sc.q, b->seqs_dev, B, b->max_nsplit, (size_t)l * sh.Hkvs * D.d, sc.ao, sh.Hs, sh.Hkvs,
                                                 D.d, D.scale, 0.0));
        }
        // (a rank's row-parallel outputs: complete, no slabs -- the all-reduce wants the sum; sums in rank order on every rank)
        FL_TRY(launch_linear(L, dt, ly.wo, sc.ao, nullptr, sc.delta, T, D.h, sh.Hs * D.d, EPI_F32, nullptr, tpr ? 1 : kMaxKSplit, &nslab));
        if (tpr) FL_TRY(oneshot(m, sh, false, sc.delta, sc.delta, T * D.h, 0));
        FL_TRY(launch_rmsnorm_add(L, dt, sc.x_res, sc.delta, ly.ln2, D.eps, sc.xn, sc.inv_rms, T, D.h, nslab, slab));
        FL_TRY(wide(ly.wgu, sc.act, 2 * sh.Ip, EPI_GATEUP));
        FL_TRY(launch_linear(L, dt, ly.wd, sc.act, nullptr, sc.delta, T, D.h, sh.Ip, EPI_F32, nullptr, tpr ? 1 : kMaxKSplit, &nslab));
        if (tpr) FL_TRY(oneshot(m, sh, false, sc.delta, sc.delta, T * D.h, 0));
    }
    FL_TRY(launch_rmsnorm_add(L, dt, sc.x_res, sc.delta, sh.norm, D.eps, sc.xn, sc.inv_rms, T, D.h, nslab, slab));
    if (tpr) {
        // every rank's [B][V / tp] block of the logits, gathered in one collective (rank-major) and laid out [B][V] for token selection
        FL_TRY(wide(sh.lm_head, b->logits_local, sh.Vs, EPI_F32));
        FL_TRY(oneshot(m, sh, true, b->logits_local, b->logits_ranks, (int64_t)B * sh.Vs, (int64_t)B * sh.Vs));
        FL_TRY(launch_unshard_logits(L, b->logits_ranks, b->logits, m->tp, B, sh.Vs));
    } else {
        FL_TRY(wide(sh.lm_head, b->logits, D.V, EPI_F32));
    }
    return launch_select_advance_batch(L, b->logits, D.V, b->seqs_dev, B, 1);
}

static int batch_step(Batch *b) {
    Model *m = b->m;
    Shard &sh = m->shards[0];
    const bool graphable = m->use_graph && !m->profiling && !b->graph_failed;
    if (graphable && b->graph) { FL_HIP(hipGraphLaunch(b->graph, sh.stream)); return FL_OK; }
    if (graphable && b->warm_steps >= 1) {
        hipGraph_t g = nullptr;
        bool ok = hipStreamBeginCapture(sh.stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            const int rc = enqueue_batch_step(b);
            const hipError_t e = hipStreamEndCapture(sh.stream, &g);
            ok = rc == FL_OK && e == hipSuccess && g != nullptr;
        }
        if (ok) ok = hipGraphInstantiate(&b->graph, g, nullptr, nullptr, 0) == hipSuccess;
        if (g) (void)hipGraphDestroy(g);
        if (ok) { FL_HIP(hipGraphLaunch(b->graph, sh.stream)); return FL_OK; }
        (void)hipGetLastError();
        b->graph_failed = true; b->graph = nullptr;
    }
    FL_TRY(enqueue_batch_step(b));
    b->warm_steps++;
    return FL_OK;
}

static int batch_check(Batch *b, const size_t *pos, size_t n_steps) {
    if (!b || !pos) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
    for (int i = 0; i < b->B; i++) FL_TRY(check_call(b->m, b->caches[i], n_steps, pos[i]));
    return FL_OK;
}

int batch_decode(Batch *b, const uint32_t *first, const size_t *pos, size_t n_steps, int64_t eos,
                 const fl_sampling *sampling, uint32_t *tokens_out, size_t *n_out) {
    if (!b) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
    std::vector<int64_t> e((size_t)b->B, eos);
    std::vector<fl_sampling> sp((size_t)b->B, sampling ? *sampling : fl_sampling{0.0, 0, 0});
    return batch_decode_each(b, first, pos, n_steps, e.data(), sp.data(), tokens_out, n_out);
}

// ... with every sequence's own EOS id and sampler (a request's temperature is its own: chat.rs:24-25; temperature < 1e-7 = ArgMax)
int batch_decode_each(Batch *b, const uint32_t *first, const size_t *pos, size_t n_steps, const int64_t *eos_each,
                      const fl_sampling *sampling_each, uint32_t *tokens_out, size_t *n_out) {
    if (!b || !first || !tokens_out || !n_out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
    const int B = b->B;
    for (int i = 0; i < B; i++) n_out[i] = 0;
    if (n_steps == 0) return FL_OK;
    FL_TRY(batch_check(b, pos, n_steps));
    Model *m = b->m;
    for (int i = 0; i < B; i++) if ((int64_t)first[i] >= m->D.V) FL_FAIL(FL_ERR_BAD_ARGUMENT, "token id %u out of range", first[i]);
    std::vector<SampleState> samplers((size_t)B);
    std::vector<int64_t> eoss((size_t)B, -1);
    for (int i = 0; i < B; i++) {
        samplers[(size_t)i] = make_sampler(sampling_each ? sampling_each + i : nullptr);
        if (eos_each) eoss[(size_t)i] = eos_each[i];
    }
    std::lock_guard<std::mutex> lock(m->mu);
    Shard &sh = m->shards[0];
    FL_HIP(hipSetDevice(sh.device));
    std::vector<uint32_t> tok(first, first + B);
    std::vector<char> finished(B, 0);
    std::vector<size_t> len0(B);
    for (int i = 0; i < B; i++) len0[i] = b->caches[i]->len;
    size_t done = 0;
    while (done < n_steps) {
        const size_t nb = std::min(n_steps - done, kBatchChunk);
        for (int i = 0; i < B; i++) {
            Cache *c = b->caches[i];
            hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(64), 0, sh.stream, c->shards[0].st, tok[i], (uint32_t)(pos[i] + done),
                               (uint32_t)(len0[i] + done), (uint32_t)(len0[i] + done), 0u, (int32_t)eoss[(size_t)i], c->shards[0].heads_done, (int)(m->D.L * sh.Hkvs),
                               c->shards[0].ss, samplers[(size_t)i], done == 0 ? 1 : 0);
            FL_HIP(hipGetLastError());
        }
        for (size_t s = 0; s < nb; s++) FL_TRY(batch_step(b));
        for (int i = 0; i < B; i++) {
            FL_HIP(hipMemcpyAsync(b->host_tokens + (size_t)i * kBatchChunk, b->caches[i]->shards[0].out_tokens, nb * 4, hipMemcpyDeviceToHost, sh.stream));
            FL_HIP(hipMemcpyAsync(b->host_states + i, b->caches[i]->shards[0].st, sizeof(StepState), hipMemcpyDeviceToHost, sh.stream));
        }
        FL_HIP(hipStreamSynchronize(sh.stream));
        FL_TRY(comm_check(m));                                    // (a tensor-parallel rank: a collective that gave up waiting for a peer)
        bool all_finished = true;
        for (int i = 0; i < B; i++) {
            if (b->host_states[i].error) FL_FAIL(FL_ERR_HIP, "device-side wait gave up (code 0x%x) in sequence %d", b->host_states[i].error, i);
            if (!finished[i]) {
                for (size_t s = 0; s < nb; s++) {
                    const uint32_t t = b->host_tokens[(size_t)i * kBatchChunk + s];
                    if (eoss[(size_t)i] >= 0 && (int64_t)t == eoss[(size_t)i]) {   // as fl_decode_greedy: the EOS forward counts, the token does not
                        finished[i] = 1;
                        b->caches[i]->len = len0[i] + done + s + 1;
                        break;
                    }
                    tokens_out[(size_t)i * n_steps + done + s] = t;
                    n_out[i] = done + s + 1;
                }
                if (!finished[i]) b->caches[i]->len = len0[i] + done + nb;
            }
            tok[i] = b->host_tokens[(size_t)i * kBatchChunk + nb - 1];
            all_finished = all_finished && finished[i];
        }
        done += nb;
        if (all_finished) break;
    }
    return FL_OK;
}

int batch_forward(Batch *b, const uint32_t *tokens, const size_t *pos, float *logits_out, uint32_t *tokens_out) {
    if (!b || !tokens) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
    FL_TRY(batch_check(b, pos, 1));
    Model *m = b->m;
    const int B = b->B;
    for (int i = 0; i < B; i++) if ((int64_t)tokens[i] >= m->D.V) FL_FAIL(FL_ERR_BAD_ARGUMENT, "token id %u out of range", tokens[i]);
    const SampleState sampler{};
    std::lock_guard<std::mutex> lock(m->mu);
    Shard &sh = m->shards[0];
    FL_HIP(hipSetDevice(sh.device));
    for (int i = 0; i < B; i++) {
        Cache *c = b->caches[i];
        hipLaunchKernelGGL(set_state_kernel, dim3(1), dim3(64), 0, sh.stream, c->shards[0].st, tokens[i], (uint32_t)pos[i], (uint32_t)c->len,
                           (uint32_t)c->len, 0u, (int32_t)-1, c->shards[0].heads_done, (int)(m->D.L * sh.Hkvs), c->shards[0].ss, sampler, 1);
        FL_HIP(hipGetLastError());
    }
    FL_TRY(batch_step(b));
    if (logits_out) FL_HIP(hipMemcpyAsync(logits_out, b->logits, (size_t)B * m->D.V * 4, hipMemcpyDeviceToHost, sh.stream));
    for (int i = 0; i < B; i++) {
        FL_HIP(hipMemcpyAsync(b->host_tokens + (size_t)i * kBatchChunk, b->caches[i]->shards[0].out_tokens, 4, hipMemcpyDeviceToHost, sh.stream));
        FL_HIP(hipMemcpyAsync(b->host_states + i, b->caches[i]->shards[0].st, sizeof(StepState), hipMemcpyDeviceToHost, sh.stream));
    }
    FL_HIP(hipStreamSynchronize(sh.stream));
    FL_TRY(comm_check(m));
    for (int i = 0; i < B; i++) {
        if (b->host_states[i].error) FL_FAIL(FL_ERR_HIP, "device-side wait gave up (code 0x%x) in sequence %d", b->host_states[i].error, i);
        b->caches[i]->len += 1;
        if (tokens_out) tokens_out[i] = b->host_tokens[(size_t)i * kBatchChunk];
    }
    return FL_OK;
}

}  // namespace fl
