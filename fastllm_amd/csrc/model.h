// model.h -- model / cache objects behind the C ABI and the forward-pass orchestration.
#pragma once
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include <rccl/rccl.h>

#include "kernels.h"

namespace fl {

struct Dims {                    // resolved config (defaults applied)
    int family = 0, qkv_bias = 0;
    int64_t h = 0, inter = 0, V = 0, L = 0, H = 0, Hkv = 0, max_pos = 0, window = -1;
    int64_t dm = 0;              // the MODEL's head_dim = hidden_size / num_attention_heads (any even value, config.rs:31-43)
    int64_t d = 0;               // the head_dim the kernels run: dm padded to 64 or 128 with zero weight rows (model.hip, build_weights)
    float eps = 0.f, scale = 0.f;
    double theta = 10000.0;
};

int resolve_config(const fl_config *cfg, Dims *out);
int tp_slice(const Dims &D, const char *name, int rank, int tp, int64_t out[4]);

struct LayerW {
    void *wqkv = nullptr;        // [(Hs+2Hkvs)*d, h]  q | k | v rows of this shard
    float *bqkv = nullptr;       // [(Hs+2Hkvs)*d] or null
    void *wo = nullptr;          // [h, Hs*d]
    void *wgu = nullptr;         // [2*Ip, h] 16-interleaved gate/up rows
    void *wd = nullptr;          // [h, Ip]
    float *ln1 = nullptr, *ln2 = nullptr;
};

struct Scratch {                 // activations of one forward chunk on one shard
    int64_t cap_T = 0;
    float *x_res = nullptr;      // [T,h] fp32 residual stream
    float *x_res2 = nullptr;     // decode only: ping-pong partner of x_res for the fused norm prologue
    float *delta = nullptr;      // [T,h] fp32 output of o_proj / down_proj (all-reduced under TP)
    void *xn = nullptr;          // [T,h] x * norm_weight (compute dtype); RMSNorm = inv_rms * xn
    float *inv_rms = nullptr;    // [T]
    float *rs_part = nullptr;    // [T, np] partial sums of squares left by the residual epilogue of a long prompt's o_proj / down_proj
    float *qkv = nullptr;        // [T,(Hs+2Hkvs)*d] fp32
    void *q = nullptr;           // [T,Hs*d]
    void *ao = nullptr;          // [T,Hs*d] attention output
    void *act = nullptr;         // [T,Ip] silu(gate)*up
    uint32_t *ids = nullptr;     // [T]
};

// One-shot collectives over peer-mapped HBM (k_comm.hip): this rank's inbox and its view of the peers'.
struct PeerComm {
    void *local = nullptr;       // hipDeviceMallocUncached: [flags, 4 KB | inbox: 2 halves x tp slots x nmax floats | LL: 2 x tp x h x 8 B]
    size_t bytes = 0;
    int64_t nmax = 0;            // floats per slot
    CommTable tab{};
    void *mapped[FL_MAX_TP] = {};     // hipIpcOpenMemHandle mappings of the peers (closed on destroy)
    uint32_t *epoch = nullptr;   // device: collectives done so far (same on every rank)
    uint32_t *err = nullptr;     // pinned host word a kernel writes when a peer never showed up
    long long timeout_ticks = 0; // wall_clock64 ticks (100 MHz)
    bool connected = false;
    // all-reduce fused into the row-parallel decode GEMVs (comm_ll.h): a region behind the inbox halves
    size_t ll_off = 0;           // byte offset of the LL region in `local` (same on every rank)
    LLTable ll{};                // host copy; `ll_dev` is what the kernels read
    LLTable *ll_dev = nullptr;
    bool ll_ok = false;          // region connected (and, with RCCL, proven against ncclAllReduce by all ranks)
    bool loopback = false;       // TK_DEBUG_TP_LOOPBACK: all entries are this rank's own inbox, the kernels play every rank's push
    bool wide_ok = true;         // messages of 16 384 floats and more over many workgroups (oneshot_wide_kernel): proved at bootstrap, like the rest
    bool shares_device = false;  // a peer lives on this same GPU (rehearsals): full-chip grids that wait for each other cannot co-reside
};

struct Shard {
    int device = 0;
    int rank = 0;                // TP rank this shard plays
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;   // prefill all-reduces of one token chunk run here while the next chunk computes
    hipEvent_t ev[8] = {};               // e0 e1 (o_proj chunk done) h0 h1 (its all-reduce done) f0 f1 (down done) g0 g1 (reduced)
    int64_t Hs = 0, Hkvs = 0, Is = 0, Ip = 0, Vs = 0, v0 = 0;
    void *embed = nullptr;       // [V,h]
    std::vector<LayerW> layers;
    float *norm = nullptr;
    void *lm_head = nullptr;     // [Vs,h]
    float *cos_tab = nullptr, *sin_tab = nullptr;   // [max_pos][d/2]
    Scratch dec, pre;
    float *logits_local = nullptr;   // [Vs]
    ArgmaxCand *amax = nullptr;      // [kMaxArgmaxCand] ArgMax candidates left by the decode step's lm_head launch (GemvArgs::amax)
    bool amax_valid = false;         // ... and whether the forward enqueued last produced them (host-side, per enqueue)
    float *logits_full = nullptr;    // [V]
    // the persistent decode engine (k_engine.hip): granule edges of one launch and the tag epoch
    unsigned long long *eng_edge[3] = {};   // o_proj deltas [h] | silu(g)*u pairs [Ip/2] | down_proj deltas [h]
    uint32_t *eng_epoch = nullptr;
    std::vector<void *> allocs;
    std::vector<void *> pre_allocs;  // the prefill scratch set: replaced (and freed) when a longer prompt arrives
    int64_t pre_bytes = 0;
    ncclComm_t comm = nullptr;
    PeerComm pc;
};

struct Model {
    std::atomic<int> refs{1};
    std::mutex mu;
    Dims D;
    fl_config cfg_resolved{};
    int dtype = FL_DTYPE_BF16;   // compute dtype
    int tp = 1, tp_mode = FL_TP_NONE;
    bool vocab_parallel = false;
    std::vector<Shard> shards;   // local shards (1 except SINGLE_PROCESS / EMULATED)
    float **emu_ptrs = nullptr;  // EMULATED: device table of per-shard buffers for the local reduce
    float *host_logits = nullptr;   // pinned staging [V]
    uint32_t *host_tokens = nullptr;
    bool use_graph = true;
    bool fused_decode = true;    // norm / RoPE / KV-append fused into the GEMV kernels
    int engine = -1;             // persistent decode engine (k_engine.hip): 0 off, 1 on, -1 automatic
    int fuse_oproj = 0;          // decode attention + o_proj in one launch (k_attn_oproj.hip): 0 never (default), 1 wherever it fits, -1 where it pays most
    StepState *host_state = nullptr;   // pinned: device step state read back for the error word
    std::vector<ProfRecord> prof;
    bool profiling = false;
    int64_t hbm_bytes = 0;
    size_t esize() const { return dtype == FL_DTYPE_BF16 ? 2 : 4; }
    ~Model();
};

struct CacheShard {
    void *k = nullptr, *v = nullptr;     // [L][Hkvs][max_seq][d]
    StepState *st = nullptr;
    SampleState *ss = nullptr;           // token selection: ArgMax or seeded temperature sampling
    float *sel_scratch = nullptr;        // [V] probabilities / cumulative weights of the sampling path
    uint32_t *out_tokens = nullptr;
    float *part_m = nullptr, *part_l = nullptr, *part_o = nullptr;
    unsigned *counters = nullptr;        // split-S arrival tickets, [Hkvs * q-groups]
    unsigned *heads_done = nullptr;      // [L][Hkv] fused attention+o_proj: split partials published per kv head since set_state
    float *ao_part = nullptr;            // ... and the partials: [Hs][ao_nsplit][d + 4] (o, m, l)
    hipGraphExec_t graph = nullptr;
    std::vector<void *> allocs;
};

constexpr size_t kOutTokensCap = 4096;

struct Cache {
    Model *m = nullptr;
    size_t max_seq = 0, len = 0;
    size_t seq_alloc = 0;        // max_seq rounded up to 32: row stride of K / column stride of V^T
    bool v_transposed = false;   // bf16 MFMA attention: value cache stored [Hkvs][d][seq_alloc]
    bool rep_attn = false;       // short cache: decode attention replicated inside the o_proj launch (k_attn_rep.hip)
    bool fuse_oproj = false;     // this cache's decode steps use the fused attention+o_proj launch
    int ao_nsplit = 0, ao_waves = 4;   // ... with this many key splits per kv head, 32 * ao_waves keys per split and step
    int nsplit = 1;
    int warm_steps = 0;          // eager decode steps done (graph is captured after the first)
    bool graph_failed = false;
    std::vector<CacheShard> shards;
    ~Cache();
};

// A fixed set of B <= 64 caches of one model decoded together: one read of the weights per step serves every sequence.
// Up to 8 rows the projections are the streaming GEMVs of k_gemv_batch.hip / k_gemv_dma.hip; beyond, the step is the
// prefill-shaped one (rmsnorm_add, short-prompt GEMM, RoPE, attention as launches of their own) at T = B.
// Single GPU, bf16, MFMA-attention shapes.
constexpr int kMaxBatch = 64;
constexpr int kMaxBatchGemv = 8;                 // rows the streaming GEMV kernels hold
struct Batch {
    Model *m = nullptr;
    std::vector<Cache *> caches;
    int B = 0, max_nsplit = 1, nks_o = 1, nks_down = 1;
    bool dma = false;                            // B >= 3: projections on the LDS-DMA ring kernel (k_gemv_dma.hip)
    bool unfused = false;                        // large B: norm / GEMM / RoPE as separate launches around the short-prompt GEMM
    bool plain = false;                          // ... all caches in the plain layout and head_dim 64 / 128: the batch kernels' plain-layout forms (one RoPE and one attention launch per layer)
    bool per_seq = false;                        // fp32 models, caches without the MFMA attention layout: embedding / RoPE / attention as launches per sequence (the single-sequence kernels on row i), the projections still once for all B rows
    Scratch sc;                                  // ... with the prefill scratch layout at T = B
    SeqRef *seqs_dev = nullptr;
    float *x_res = nullptr, *x_res2 = nullptr;   // [B][h]
    float *delta = nullptr;                      // [max(nks_o, nks_down)][B][h]
    void *q = nullptr, *ao = nullptr;            // [B][H*d]
    void *act = nullptr;                         // [B][Ip]
    float *logits = nullptr;                     // [B][V]
    float *logits_local = nullptr, *logits_ranks = nullptr;   // tensor-parallel rank: its [B][Vs] block, and the gathered [tp][B][Vs]
    uint32_t *host_tokens = nullptr;             // pinned [B][kBatchChunk]
    StepState *host_states = nullptr;            // pinned [B]
    hipGraphExec_t graph = nullptr;
    bool graph_failed = false;
    int warm_steps = 0;
    std::vector<void *> allocs;
    ~Batch();
};
constexpr size_t kBatchChunk = 256;              // decode steps enqueued between two looks at the tokens

int batch_create(Model *m, Cache *const *caches, size_t B, Batch **out);
int batch_replace(Batch *b, size_t slot, Cache *c);     // fl_batch_replace: another stream's cache takes sequence `slot`'s place
// one step for every sequence: tokens[b] at RoPE offset pos[b]; logits_out [B][V] (host) or null
int batch_forward(Batch *b, const uint32_t *tokens, const size_t *pos, float *logits_out, uint32_t *tokens_out);
// n_steps greedy / sampled steps; tokens_out [B][n_steps], n_out [B] (stops counting a sequence at its EOS)
int batch_decode(Batch *b, const uint32_t *first, const size_t *pos, size_t n_steps, int64_t eos,
                 const fl_sampling *sampling, uint32_t *tokens_out, size_t *n_out);
int batch_decode_each(Batch *b, const uint32_t *first, const size_t *pos, size_t n_steps, const int64_t *eos_each,
                      const fl_sampling *sampling_each, uint32_t *tokens_out, size_t *n_out);

int model_create(const fl_config *cfg, const fl_tensor *tensors, size_t n, int compute_dtype,
                 const fl_parallel *par, Model **out);
int cache_create(Model *m, size_t max_seq, Cache **out);
// FL_TP_MULTI_PROCESS: export this rank's inbox / map the peers' (handles: tp x FL_IPC_HANDLE_BYTES in rank order)
int comm_ipc_export(Model *m, void *handle_out);
int comm_ipc_connect(Model *m, const void *handles);
bool fused_all_reduce_ready(const Model *m);   // decode all-reduces ride in the GEMV epilogues (comm_ll.h)

// shared by model.hip and comm.hip
#define FL_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { \
    ::fl::set_error("RCCL error %s at %s:%d (%s)", ncclGetErrorString(r_), __FILE__, __LINE__, #expr); \
    return FL_ERR_RCCL; } } while (0)
int env_int(const char *name, int dflt);
void debug_inject(const char *site);     // FL_DEBUG_THROW fault injection (tests of the ABI exception barrier)
Launcher make_launcher(Model *m, Shard &sh);
// comm.hip: inbox / LL region of one shard, its table entries, and the group-level steps
int comm_alloc(Model *m, Shard &sh);
void comm_inbox_release(int device, size_t bytes, void *p);   // a destroyed shard's uncached inbox goes to a free list, never back to the runtime (comm.hip)
void comm_forget(void *local);                         // a destroyed shard's inbox leaves the table of handles exported by this process
void comm_set_entry(PeerComm &pc, int r, void *base);
int comm_ll_publish(Model *m, Shard &sh);              // every entry is set: hand the fused all-reduce's table to the device
int comm_connect_loopback(Model *m);                   // TK_DEBUG_TP_LOOPBACK: every entry is this rank's own inbox (timing tool)
int comm_bootstrap_over_rccl(Model *m);                // all-gather the handles through RCCL, connect, self-test, vote
int comm_probe(Model *m, int form, int64_t n, int iters, double *us_per_call);   // fl_comm_probe
int comm_selftest(Model *m, int64_t n, int *ok);                                  // fl_comm_selftest
int comm_check(Model *m);                              // a kernel gave up waiting for a peer -> FL_ERR_RCCL
// n floats in chunks of at most the inbox size; reduce: out = sum over ranks (in == out allowed); gather: out[r * out_stride + i] = in_r[i]
int oneshot(Model *m, Shard &sh, bool gather, const float *in, float *out, int64_t n, int64_t out_stride, hipStream_t on = nullptr);
// mode: 0 = logits to host, 1 = argmax token to host
// device token-selection state for a LogitsProcessor::new(seed, Some(temperature), None); null: ArgMax
SampleState make_sampler(const fl_sampling *sampling);
// sampling == null: ArgMax
int forward(Model *m, Cache *c, const uint32_t *ids, size_t T, size_t pos, float *logits_out, uint32_t *token_out,
            const fl_sampling *sampling = nullptr);
int decode_greedy(Model *m, Cache *c, uint32_t first, size_t pos, size_t n_steps, int64_t eos,
                  uint32_t *tokens_out, size_t *n_out, const fl_sampling *sampling = nullptr);

// most K slices (fp32 slabs summed by the next launch) a row-parallel projection may use at T tokens (model.hip)
int ksplit_cap(int64_t T);

}  // namespace fl
