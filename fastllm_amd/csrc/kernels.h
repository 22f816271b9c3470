// kernels.h -- host-callable launch wrappers of the gfx950 kernels.
#pragma once
#include <vector>

#include "common.h"

namespace fl {

// Kernel classes for the measurement hooks (fl_profile_*): one class per kernel symbol family.
enum KernelClass {
    KC_EMBED = 0, KC_RMSNORM, KC_GEMV, KC_GEMM_MFMA, KC_GEMM_GENERIC, KC_ROPE_KV, KC_ATTN_DECODE,
    KC_ATTN_COMBINE, KC_ATTN_PREFILL, KC_ARGMAX, KC_REDUCE, KC_CONVERT, KC_ATTN_OPROJ, KC_COMM, KC_COUNT
};
const char *kernel_class_name(int kc);

struct ProfRecord { int kc; double bytes, flops; hipEvent_t e0, e1; char tag[32]; };

// 1/rms of the rows a residual epilogue (EPI_RESID) left, taken by the CONSUMING projection itself from the partial sums of squares
// -- row t: 1 / sqrt(sum_i part[t][i] / h + eps), summed left to right -- instead of an rms_finalize launch in between.  Honoured by
// the 256 x 256 kernels' plain launches and by the 128 x 256 kernel (gemm_takes_rs_parts tells); everything else ignores it.
struct RsParts { const float *part = nullptr; int np = 0; float eps = 0.f, inv_h = 0.f; };

// Launch context: the stream a kernel goes to and, while profiling, where its event pair is kept.
struct Launcher {
    hipStream_t stream = nullptr;
    RsParts rsp;                               // non-null part: the next projection's row scales come from these partial sums
    std::vector<ProfRecord> *prof = nullptr;   // non-null: bracket every launch with HIP events
    const char *tag = "";                      // optional sub-class label for the profile (shape, variant)
    int tp = 1;                                // tensor-parallel degree of the model this launch belongs to (> 1: a rank's shard shapes -- launch_linear plans for those)

    template <typename... KArgs, typename... Args>
    int launch(int kc, double bytes, double flops, void (*kernel)(KArgs...), dim3 grid, dim3 block,
               size_t lds, Args... args) {
        if (grid.x == 0 || grid.y == 0 || grid.z == 0) return FL_OK;
        if (prof) {
            ProfRecord r{kc, bytes, flops, nullptr, nullptr, {0}};
            snprintf(r.tag, sizeof r.tag, "%s", tag ? tag : "");
            FL_HIP(hipEventCreate(&r.e0));
            FL_HIP(hipEventCreate(&r.e1));
            hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, stream, r.e0, r.e1, 0, static_cast<KArgs>(args)...);
            prof->push_back(r);
        } else {
            hipLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, stream, static_cast<KArgs>(args)...);
        }
        FL_HIP(hipGetLastError());
        return FL_OK;
    }
};

enum { EPI_F32 = 0, EPI_GATEUP = 1, EPI_QKV_ROPE = 2, EPI_RESID = 3 };
// EPI_RESID (256x256 prefill GEMM only): the o_proj / down_proj epilogue takes over the residual add and the next
// RMSNorm's first pass -- h += y (fp32, in place), xn = (h + y) * w in the compute dtype, and partial sums of squares of
// h + y per (row, column tile, wave column) that rms_finalize turns into 1/rms -- instead of a y round trip through
// HBM and an rmsnorm_add launch.
struct ResidEpi {
    float *h = nullptr;            // [T][N] residual stream
    const float *w = nullptr;      // [N] weight of the NEXT RMSNorm
    void *xn = nullptr;            // [T][N] bf16
    float *part = nullptr;         // [T][np]
    int np = 0;
};
enum { PRO_X = 0, PRO_NORM = 1 };

struct LLTable;
// Arguments of the decode weight-streaming kernel (k_gemv.hip), passed by value as kernarg.
struct ArgmaxCand { float v; int i; };   // [0]: {unused, count}; [1 + workgroup]: its best row
constexpr int kMaxArgmaxCand = 1024;

struct GemvArgs {
    const void *W = nullptr;        // [N,K] compute dtype
    const void *x = nullptr;        // PRO_X: [K] compute dtype
    const float *x_scale = nullptr; // PRO_X: optional scalar applied to the accumulator (1/rms)
    const float *bias = nullptr;    // [N] or null
    void *out = nullptr;            // EPI_F32: float[N]; EPI_GATEUP: XT[N/2]
    int N = 0, K = 0, epi = EPI_F32, pro = PRO_X;
    // PRO_NORM: x = rmsnorm(x_in + delta) * norm_w, or of the embedding row of st->token
    const float *x_in = nullptr, *delta = nullptr, *norm_w = nullptr;
    int delta_nslab = 1;            // delta is the sum of this many [K] vectors, K floats apart (the fused attention + o_proj launch leaves one per kv head)
    float eps = 0.f;
    float *x_out = nullptr;         // updated residual, written by workgroup 0 (must differ from x_in)
    const void *embed = nullptr;    // [V,K] compute dtype, or null
    const StepState *st = nullptr;
    // EPI_QKV_ROPE: rows are q | k | v heads of width d; RoPE + KV append fused
    const float *cos_tab = nullptr, *sin_tab = nullptr;
    void *q_out = nullptr, *k_cache = nullptr, *v_cache = nullptr;
    int H = 0, Hkv = 0, d = 0, max_seq = 0, max_pos = 0;
    int v_ld = 0;                   // > 0: value cache is transposed [Hkv][d][v_ld] (MFMA attention layout)
    // EPI_F32 of a row-parallel projection in a tensor-parallel group: out = sum over ranks, exchanged in the epilogue
    const LLTable *ll = nullptr;    // device-resident; null = plain local output
    int ll_slot = 0;                // 1 .. LLTable::slots: which all-reduce of the decode step this is
    // EPI_F32 (the lm_head launch of a decode step): every workgroup also leaves the ArgMax of the rows it computed
    // ((value, index) pairs; ties -> the larger index, as argmax_last), so that token selection reads gridDim.x candidates
    // instead of the whole vocabulary; amax[0] is the candidate count.  null = not wanted.
    ArgmaxCand *amax = nullptr;
};
int launch_gemv(Launcher &L, int dtype, const GemvArgs &a);
// bytes of W that workgroup b of the plain-epilogue launch (no norm prologue) of an N x K matrix reads as one contiguous piece
// ([b chunk, (b + 1) chunk), then every grid-th piece): what a prefetcher needs to know to warm the right XCD's L2
int64_t gemv_owner_chunk(int dtype, int64_t N, int64_t K);
bool gemv_leaves_candidates(int dtype, const GemvArgs &a);   // the grid launch_gemv would pick fits the candidate buffer
bool gemv_norm_supported(int dtype, int64_t N, int64_t K);
void gemv_set_tuning(int blocks, int waves);   // fl_tune "gemv_blocks" / "gemv_waves": force the grid / waves per workgroup (0 automatic, -1 keep)

// ---- the persistent decode engine (k_engine.hip): a chain of projections in one launch ---------------------------
constexpr int ENG_GATHER_WAVES = 4, ENG_STREAM_WAVES = 8, ENG_MAX_OPS = 4;
enum { ENG_IN_X = 0, ENG_IN_NORM = 1, ENG_IN_ACT = 2 };                 // plain bf16 vector of the previous launch | fp32 delta edge + residual + RMSNorm weight | packed silu(g)*u edge
enum { ENG_OUT_EDGE_F32 = 0, ENG_OUT_EDGE_ACT = 1, ENG_OUT_QKV = 2, ENG_OUT_LOGITS = 3 };
struct EngOp {
    const void *W = nullptr;            // [N,K] bf16
    int N = 0, K = 0, in = ENG_IN_X, out = ENG_OUT_EDGE_F32;
    int R = 2;                          // rows per wave item (launch_engine: 1 for row-parallel ops with few rows per CU)
    const void *x = nullptr;            // ENG_IN_X: bf16 [K]
    const float *norm_w = nullptr;      // ENG_IN_NORM: weight of the RMSNorm in front of this op
    const unsigned long long *in_edge = nullptr;   // granules {value, tag} written by the previous op of this launch
    unsigned long long *out_edge = nullptr;
    float *res_out = nullptr;           // ENG_IN_NORM: workgroup 0 also leaves the updated residual stream here (the next launch's x_res_in)
    void *dst = nullptr;                // ENG_OUT_LOGITS: float [N]
    const float *bias = nullptr;        // ENG_OUT_QKV / ENG_OUT_LOGITS
    int tag_in = 0, tag_out = 0;        // 1..255, unique per (launch of the step, edge)
};
struct EngArgs {
    EngOp op[ENG_MAX_OPS];
    int nops = 0, h = 0;
    const float *x_res_in = nullptr;    // [h] residual stream in front of the first op's layer (null: no op of the chain norms)
    float eps = 0.f;
    const StepState *st = nullptr;
    StepState *st_rw = nullptr;         // error word of a wait that gave up
    const uint32_t *epoch = nullptr;    // device word advanced once per decode step (select_advance): tags never repeat
    long long timeout_ticks = 0;        // wall_clock64 ticks (100 MHz)
    // ENG_OUT_QKV: RoPE + KV append as in GemvArgs
    const float *cos_tab = nullptr, *sin_tab = nullptr;
    void *q_out = nullptr, *k_cache = nullptr, *v_cache = nullptr;
    int H = 0, Hkv = 0, d = 0, max_seq = 0, max_pos = 0, v_ld = 0;
    ArgmaxCand *amax = nullptr;         // ENG_OUT_LOGITS: one ArgMax candidate per workgroup (as GemvArgs::amax)
    unsigned long long *stamps = nullptr;   // diagnostics: [workgroup][32] wall_clock64 stamps (FL_ENGINE_STAMPS)
    int grid = 0;                       // 0: one workgroup per CU
    int xs0_bytes = 0, xs1_bytes = 0;   // (filled by launch_engine)
    int pf_blocks = 1;                  // 8-KiB blocks a streamer requests ahead of an op's input (FL_ENGINE_PF)
    int gather_delay = 0;               // s_sleep(1) rounds between "this CU's streamers are done" and the first sweep (FL_ENGINE_DELAY)
};
bool engine_shape_ok(int64_t h, int64_t Hd, int64_t I, int64_t N_last);
int launch_engine(Launcher &L, const EngArgs &a);
void engine_set_grid(int workgroups);   // fl_tune("engine_grid"): 0 = one per CU

// ---- batched decode (k_gemv_batch.hip): B <= 8 sequences share one read of the weights ------------
// Per-sequence device state and buffers of a batch member (a view of its cache).
struct SeqRef {
    StepState *st;
    SampleState *ss;
    void *k, *v;                    // cache bases [L][Hkv][seq_alloc][d] / V^T [L][Hkv][d][seq_alloc]
    float *part_m, *part_l, *part_o;
    unsigned *counters;
    uint32_t *out_tokens;
    float *sel_scratch;
    int seq_alloc, nsplit;
};
struct GemvBatchArgs {
    const void *W = nullptr;        // [N,K] bf16
    const float *bias = nullptr;
    void *out = nullptr;            // EPI_F32: float [nks][B][N]; EPI_GATEUP: bf16 [B][N/2]
    int N = 0, K = 0, epi = EPI_F32, pro = PRO_X, B = 1, nks = 1;
    const void *x = nullptr;        // PRO_X: bf16 [B][K]
    const float *x_scale = nullptr; // PRO_X (k_gemv_dma.hip): optional [B] factors applied to the accumulators (1/rms of a separate norm)
    // PRO_NORM: x[b] = rmsnorm(x_in[b] + sum_s delta[s][b]) * norm_w, or of the embedding row of seqs[b].st->token
    const float *x_in = nullptr, *delta = nullptr, *norm_w = nullptr;
    int delta_nslab = 1;            // delta is the sum of this many [K] vectors, K floats apart (the fused attention + o_proj launch leaves one per kv head)
    int n_slab = 1; long long slab_stride = 0;
    float eps = 0.f;
    float *x_out = nullptr;         // [B][K] updated residual (must differ from x_in)
    const void *embed = nullptr;
    const SeqRef *seqs = nullptr;   // device array [B]
    // EPI_QKV_ROPE
    const float *cos_tab = nullptr, *sin_tab = nullptr;
    void *q_out = nullptr;          // bf16 [B][H*d]
    size_t kv_layer_off = 0;        // layer * Hkv * d: multiplied by the sequence's seq_alloc inside
    int H = 0, Hkv = 0, d = 0, max_pos = 0;
};
int gemv_batch_ksplit(int B, int64_t K, int64_t N, int epi);
int launch_gemv_batch(Launcher &L, const GemvBatchArgs &a);
// the same projections with the weights streamed by LDS-DMA into wave-private rings (k_gemv_dma.hip; 3 <= B <= 32 rows: a batch's streams or a short prompt's tokens)
bool gemv_dma_supported(int B, int64_t N, int64_t K, int epi, int d);
int gemv_dma_ksplit(int64_t K, int64_t N, int epi);
int launch_gemv_dma(Launcher &L, const GemvBatchArgs &a);
// q / out bf16 [B][H*d]; kv_layer_off = layer * Hkv * d (times each sequence's seq_alloc inside)
int launch_attn_decode_mfma_batch(Launcher &L, const void *q, const SeqRef *seqs_dev, int B, int max_nsplit, size_t kv_layer_off,
                                  void *out, int64_t H, int64_t Hkv, int64_t d, float scale, double kv_bytes_hint);
int launch_embed_batch(Launcher &L, const void *E, const SeqRef *seqs_dev, float *x_res, int B, int64_t h, int dtype = FL_DTYPE_BF16);
// the plain-layout batch attention (k_attn.hip): fp32 models, bf16 caches outside the MFMA attention layout; head_dim 64 / 128
bool attn_decode_batch_supported(int64_t d);
int launch_attn_decode_batch(Launcher &L, int dtype, const void *q, const SeqRef *seqs_dev, int B, int max_nsplit, size_t kv_layer_off, void *out,
                             int64_t H, int64_t Hkv, int64_t d, float scale);
// n_slab: K slices of the QKV projection ([n_slab][B][(H+2Hkv)*d] fp32, summed here); bias: added here (a K-sliced
// projection cannot add it itself) or null
int launch_rope_kv_batch(Launcher &L, const float *qkv, const SeqRef *seqs_dev, const float *cos_tab, const float *sin_tab,
                         int64_t max_pos, void *q_out, size_t kv_layer_off, int B, int64_t H, int64_t Hkv, int64_t d, int n_slab = 1,
                         const float *bias = nullptr, int dtype = FL_DTYPE_BF16, bool v_transposed = true);
// logits fp32 [B][V] -> every sequence's token / step state
int launch_unshard_logits(Launcher &L, const float *in, float *out, int tp, int B, int64_t Vs);   // [tp][B][Vs] -> [B][tp * Vs]
int launch_select_advance_batch(Launcher &L, const float *logits, int64_t V, const SeqRef *seqs_dev, int B, int advance);

// dst row of gate/up pair q in the 16-interleaved fused layout: 16 gate rows then 16 up rows
__host__ __device__ inline int64_t gateup_row(int64_t q, int is_up) { return (q / 16) * 32 + (q % 16) + (is_up ? 16 : 0); }

// ---- projections ---------------------------------------------------------------------------
// y = x[T,K] . W[N,K]^T (+bias).  EPI_F32: y fp32 [T,N].  EPI_GATEUP: W is the 16-interleaved
// gate/up matrix (N = 2*I rows), y is XT [T, I] = silu(gate)*up.
// dtype FL_DTYPE_BF16: W and x (and gate-up y) bf16; FL_DTYPE_F32: all fp32.
// row_scale (optional, fp32 [T]): y[t,:] = row_scale[t] * (x[t,:] . W^T) (+bias) -- the 1/rms factor of
// a preceding RMSNorm, applied after the dot product (see launch_rmsnorm_add).
// max_split > 1 allows split-K: y then holds *n_split_out fp32 slabs of [T,N] that the consumer sums
// (launch_rmsnorm_add does); *n_split_out is always written when the pointer is given.
int launch_linear(Launcher &L, int dtype, const void *W, const void *x, const float *bias, void *y,
                  int64_t T, int64_t N, int64_t K, int epi, const float *row_scale = nullptr,
                  int max_split = 1, int *n_split_out = nullptr);
// ldc: row stride of y in elements of the FULL output width (0 = N): a launch may cover a column range of a wider matrix
bool gemm_takes_rs_parts(int dtype, int64_t T, int64_t N, int64_t K, int epi, int max_split);   // would launch_linear's kernel for this shape honour Launcher::rsp?
bool gemm_resid_supported(int dtype, int64_t T, int64_t N, int64_t K, int max_split);   // k_gemm_mfma.hip: would launch_gemm_resid take this shape?
int gemm_resid_partials(int64_t N);                                                       // partial sums per row (np)
int launch_gemm_resid(Launcher &L, const void *W, const void *x, int64_t T, int64_t N, int64_t K, const ResidEpi &re);
int launch_rms_finalize(Launcher &L, const float *part, int np, float eps, float *inv_rms, int64_t T, int64_t h);
// A kernel that reads its row scales as a VECTOR was picked although the caller left them as a residual epilogue's partial sums
// (Launcher::rsp): the plan (gemm_takes_rs_parts) and the launch disagree -- a switch changed in between, or the two rules drifted.
// Finish the sums into the vector the caller passed as row_scale and go on, instead of failing a forward half way through its layers.
int rs_parts_to_vector(Launcher &L, const float *row_scale, int64_t T);
int64_t gemm_8p_workspace_bytes(hipStream_t stream);   // stream-K workspace held for a stream on the current device
void gemm_8p_release_stream(hipStream_t stream);   // frees the stream-K workspace of a stream that is about to be destroyed
int launch_gemm_8p(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                   int epi, const float *row_scale, int ksplit, int64_t ldc = 0, bool streamk = false, const ResidEpi *resid = nullptr);
// EPI_QKV_ROPE (k_gemm_h4.hip): the QKV projection's epilogue applies the row scale and bias, rotates q / k and appends k / v to the
// cache -- what launch_rope_kv does behind a plain fp32 output
struct RopeEpi {
    int on = 0;
    const StepState *st = nullptr;
    const float *cos_tab = nullptr, *sin_tab = nullptr;
    int max_pos = 0;
    void *q_out = nullptr, *k_cache = nullptr, *v_cache = nullptr;     // bf16: q [T][H*d]; K [Hkv][max_seq][d]; V [Hkv][d][max_seq] (transposed) or as K
    int H = 0, Hkv = 0, d = 0, max_seq = 0, v_transposed = 0;
    int col_base = 0;                  // first column of this launch in the whole QKV matrix (a column-peeled projection's tail launch)
    // decode batch (k_gemm_skf.hip only): row t is sequence t's single new token -- its RoPE position, KV slot and caches come from
    // seqs[t] (st->pos, st->len, k / v + kv_layer_off * seq_alloc; V transposed), not from st / k_cache / v_cache
    const struct SeqRef *seqs = nullptr;
    size_t kv_layer_off = 0;
};
// 128 x 256 tile, K slices summed inside the launch (k_gemm_h4.hip): mid-size prompts
constexpr int H4_MAXS = 4;             // K slices at most
struct H4Space {
    float *part;                       // [tile][slice][128 x 256] fp32 partial accumulators, lane-major
    unsigned *ctr;                     // [2 sets][tile]{slices published, flag word: abandoned blocks' bits | CLOSED}: alternate launches use alternate sets and zero the other
};
bool gemm_h4_supported(int64_t T, int64_t N, int64_t K, int ksplit);
int gemm_h4_plan(int64_t T, int64_t N, int64_t K, int epi);
int gemm_h4_plan_whole(int64_t T, int64_t N, int64_t K, int epi);   // ... for a caller that needs one complete output (no slabs): a tensor-parallel rank's projections
// 256 x 224 four-wave tile (k_gemm_w14.hip): fp32 / gate-up epilogues, N whole 224-column tiles
// fp32 operands on the matrix cores (k_gemm_f32.hip): the fp32 mode's prompt GEMM, bit-identical to gemm_generic_kernel's fmaf chains
bool gemm_f32_mfma_supported(int64_t T, int64_t N, int64_t K);
// fp32, 2-64 token rows: a weight stream with the x rows in LDS (k_gemm_f32.hip)
bool gemv_f32_rows_supported(int64_t T, int64_t N, int64_t K, int epi);
int launch_gemv_f32_rows(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K, int epi,
                         const float *row_scale);
int launch_gemm_f32_mfma(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                         int epi, const float *row_scale);
bool gemm_w14_plan(int64_t T, int64_t N, int64_t K, int epi);
// the four-wave 256 x 256 kernel with the RoPE / bias / KV-append epilogue (k_gemm_8p.hip): a long prompt's QKV projection, whole K
bool gemm_4w_rule(int64_t T, int64_t N, int64_t K, int64_t ksteps, bool streamk);
bool gemm_4w_rope_supported(int64_t T, int64_t N, int64_t K);
int launch_gemm_4w_rope(Launcher &L, const void *W, const void *x, const float *bias, int64_t T, int64_t N, int64_t K, const float *row_scale,
                        const RopeEpi &rope);
// the launches of a long prompt's QKV projection with that epilogue (k_gemm_mfma.hip): one plain grid, or whole rounds + peeled tail columns
// on the 128 x 256 kernel -- false where launch_linear would cut K into slabs or run stream-K (rope_kv_append then sums / rotates)
bool gemm_qkv_rope_long_plan(int64_t T, int64_t N, int64_t K, int max_split);
int launch_gemm_qkv_rope_long(Launcher &L, const void *W, const void *x, const float *bias, int64_t T, int64_t N, int64_t K, const float *row_scale,
                              const RopeEpi &rope, int max_split);
int launch_gemm_w14(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                    int epi, const float *row_scale, int64_t ldc = 0);
int gemm_h4_tail_slices(int64_t T, int64_t N, int64_t K);     // a peeled GEMM's tail columns on this kernel: K slices, or 0 = the stream-K launch + fix-up   // K slices the kernel would run this shape in; 0: another kernel takes it
int launch_gemm_h4(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                   int epi, const float *row_scale, int ksplit, int64_t ldc = 0, const ResidEpi *resid = nullptr, const RopeEpi *rope = nullptr);
int64_t gemm_h4_workspace_bytes(hipStream_t stream);
// short prompts and decode batches (2-128 rows, k_gemm_skf.hip): the weight-streaming GEMM with its K slices met inside the launch and the
// fp32 / gate-up / residual + norm / RoPE + KV-append epilogues; d: head_dim of the RoPE epilogue (0 otherwise)
int gemm_skf_plan(int64_t T, int64_t N, int64_t K, int epi, int d = 0);
int launch_gemm_skf(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K, int epi,
                    const float *row_scale, int ksplit, const ResidEpi *resid = nullptr, const RopeEpi *rope = nullptr);
void gemm_skf_release_stream(hipStream_t stream);
void gemm_h4_release_stream(hipStream_t stream);   // before the owner destroys the stream
bool gemm_skinny_supported(int64_t T, int64_t N, int64_t K);
int gemm_skinny_ksplit(int64_t T, int64_t N, int64_t K, int epi, int max_split);
int launch_gemm_skinny(Launcher &L, const void *W, const void *x, const float *bias, void *y, int64_t T, int64_t N, int64_t K,
                       int epi, const float *row_scale, int ksplit);
// cheap capability probes used by tests / DESIGN numbers
bool gemv_supported(int dtype, int64_t N, int64_t K);
bool gemm_mfma_supported(int dtype, int64_t T, int64_t N, int64_t K);

// ---- small ops -----------------------------------------------------------------------------
// x_res[t,:] = E[ids[t],:]; ids == nullptr: single token read from st->token
int launch_embed(Launcher &L, int dtype, const void *E, const uint32_t *ids, const StepState *st,
                 float *x_res, int64_t T, int64_t h);
// x_res += delta (if delta); xs = x_res * w (compute dtype); inv_rms[t] = 1/sqrt(mean(x_res^2)+eps).
// RMSNorm(x) * w == inv_rms * xs: the scalar is applied by the consuming projection's epilogue, so
// the bf16 rounding point (xs) is the same in the prefill and the fused decode path.
// delta may be n_slab split-K slabs, slab s at delta + s * slab_stride floats.
int launch_rmsnorm_add(Launcher &L, int dtype, float *x_res, const float *delta, const float *w,
                       float eps, void *xs, float *inv_rms, int64_t T, int64_t h, int n_slab = 1,
                       int64_t slab_stride = 0);
// RoPE(q,k) + KV append.  qkv fp32 [T, (H+2Hkv)*d]; q_out XT [T,H*d]; caches XT [Hkv][max_seq][d]
// v_transposed: value cache laid out [Hkv][d][max_seq] instead of [Hkv][max_seq][d]
int launch_rope_kv(Launcher &L, int dtype, const float *qkv, const StepState *st, const float *cos_tab,
                   const float *sin_tab, int64_t max_pos, void *q_out, void *k_cache, void *v_cache,
                   int64_t T, int64_t H, int64_t Hkv, int64_t d, int64_t max_seq, bool v_transposed, int nslab = 1,
                   const float *bias = nullptr);   // bias: the QKV projection's, when it ran in K slices (added after the slabs are summed)
// (nslab > 1: qkv is nslab split-K slabs of [T][(H+2Hkv)*d], summed here in slab order)
// logits[V] -> st->token by ArgMax (ties: last max index) or, when ss->on, by temperature sampling with
// the seeded ChaCha12 stream; out_tokens[st->step] = token; advances pos/len/step
int launch_select_advance(Launcher &L, const float *logits, int64_t V, StepState *st, SampleState *ss, float *scratch /* [V] */,
                          uint32_t *out_tokens, int advance,
                          const ArgmaxCand *cand = nullptr /* the lm_head launch's candidates: ArgMax without reading the vocabulary again */,
                          uint32_t *epoch_bump = nullptr /* the decode engine's tag epoch: += 1 per forward */);
// dst[i] = sum_s src[s][i] for n floats, written to every src (emulated all-reduce)
int launch_reduce_shards(Launcher &L, float *const *bufs_dev, int nshards, int64_t n);

// ---- one-shot collectives over peer-mapped HBM (k_comm.hip) -----------------------------------
constexpr int FL_MAX_TP = 8;
// Per-rank view of the tp inboxes: entry r is rank r's inbox / flag array as mapped in THIS process
// (own allocation for r == rank, hipIpcOpenMemHandle / peer pointer otherwise).
struct CommTable {
    float *inbox[FL_MAX_TP]; uint32_t *flags[FL_MAX_TP];
    int loop = 0;        // 1: LOOPBACK (tools/tp_rank_loopback.py, TK_DEBUG_TP_LOOPBACK): every entry is this rank's own inbox and the kernel plays all
                         // tp ranks' pushes itself -- one rank's step timed with its exchange in place, results meaningless
};
int launch_oneshot(Launcher &L, bool gather, const float *in, float *out, const CommTable &tab, int rank, int tp,
                   int64_t n, int64_t nmax, int64_t out_stride, uint32_t *epoch_ctr, uint32_t *err, long long timeout_ticks,
                   uint32_t *abort_flag = nullptr, uint32_t *done_ctr = nullptr, int max_wgs = 0);     // done_ctr: a zeroed device word, max_wgs > 1 -> large messages go over up to max_wgs workgroups

// All-reduce fused into the epilogue of a row-parallel decode GEMV ("LL" protocol: a value and the epoch that
// validates it travel in ONE 8-byte store, so there is no fence, no flag round and no separate kernel).  Every
// wave pushes its rows' partial sums into slot [rank] of every peer's LL region, polls the tp slots of its own
// region for the same rows and adds them in rank order -- bit-identical to oneshot_kernel's sum.
// Region layout (uint64 {lo: fp32 bits, hi: epoch}): [2 halves by epoch parity][tp source ranks][n rows].
// epoch = *epoch_ctr * slots + slot: the one-shot counter moves at least once per decode step (logits gather).
struct LLTable {
    uint64_t *peer[FL_MAX_TP];      // rank r's LL region as mapped here
    const uint32_t *epoch_ctr;      // the one-shot collectives' device counter
    uint32_t *err;                  // pinned host word
    uint32_t *abort_flag;           // device word, set with err: later waits of a broken step give up after ~1 ms
    long long timeout_ticks;
    int rank, tp, n, slots;
    int loop;                       // 1: loopback (see CommTable::loop): lane p pushes into slot [p] of its own region
};
// stand-alone exerciser of ll_allreduce_rows (bootstrap self-test, tests): out[i] = sum_r in_r[i], i < n
int launch_ll_allreduce(Launcher &L, const LLTable *ll_dev, int slot, const float *in, float *out, int64_t n);

// ---- attention -----------------------------------------------------------------------------
struct AttnScratch {
    float *part_m, *part_l, *part_o; unsigned *counters; int nsplit; int64_t kv_len_hint;
    // decode (k_attn_mfma.hip): bytes the NEXT launch will stream (the layer's o_proj weights), touched by extra workgroups of
    // this launch while its own few MB leave HBM idle -- they are then served from the Infinity Cache
    // (prefetch_chunk: the bytes one workgroup of that launch reads contiguously, gemv_owner_chunk)
    const void *prefetch = nullptr; int64_t prefetch_bytes = 0, prefetch_chunk = 0, prefetch_row = 0;   // (row: bytes of one matrix row)
};
// decode: one query token over len+1 cached keys, no mask (App. A.5)
int launch_attn_decode(Launcher &L, int dtype, const void *q, const void *k_cache, const void *v_cache,
                       const StepState *st, void *out, const AttnScratch &sc, int64_t H, int64_t Hkv,
                       int64_t d, int64_t max_seq, float scale);
// prefill: T queries; key kj visible to query t iff kj < len, or (j=kj-len) j<=t and j+window>=t
int launch_attn_prefill(Launcher &L, int dtype, const void *q, const void *k_cache, const void *v_cache,
                        const StepState *st, void *out, int64_t T, int64_t H, int64_t Hkv, int64_t d,
                        int64_t max_seq, float scale, int64_t window);

// bf16 MFMA attention (k_attn_mfma.hip): needs the transposed value cache [Hkv][d][seq_alloc], seq_alloc % 32 == 0
bool attn_mfma_supported(int dtype, int64_t H, int64_t Hkv, int64_t d);
int launch_attn_decode_mfma(Launcher &L, const void *q, const void *k_cache, const void *v_cache_T, const StepState *st,
                            void *out, const AttnScratch &sc, int64_t H, int64_t Hkv, int64_t d, int64_t seq_alloc,
                            float scale);
int launch_attn_prefill_mfma(Launcher &L, const void *q, const void *k_cache, const void *v_cache_T, const StepState *st,
                             void *out, int64_t T, int64_t H, int64_t Hkv, int64_t d, int64_t seq_alloc, float scale,
                             int64_t window);
// test hook (fl_op_attention): 0 = pick by prompt length, 2 = the 16-row kernel, 3 = the 32-row kernel
void attn_prefill_force(int which);

// decode attention + o_proj in one launch (k_attn_oproj.hip): W_o is pulled into LDS while attention runs; the output is one
// partial vector per kv head
bool attn_oproj_plan(int64_t H, int64_t Hkv, int64_t d, int64_t h, int nsplit, int cus, int *n_blocks, int *rows_attn,
                     int *rows_other, size_t *lds_bytes);
int launch_attn_oproj(Launcher &L, const void *q, const void *k_cache, const void *v_cache_T, const StepState *st,
                      StepState *st_rw, float *partials /* [H][nsplit][d + 4] */, unsigned *done /* [Hkv] words of this layer */,
                      int nsplit, int attn_waves /* 1..8: 32 keys each per split and step */, int64_t kv_len_hint, const void *Wo, float *slabs /* [Hkv][h] */, int64_t H, int64_t Hkv, int64_t d,
                      int64_t h, int64_t seq_alloc, float scale);

// decode attention + o_proj in one launch for SHORT caches (k_attn_rep.hip): every workgroup computes the attention of all heads
// itself (K / V from L2) while its rows of W_o stream in; out = W_o . attention (fp32 [N]), or summed over the ranks (ll)
struct AttnRepArgs {
    const void *q = nullptr, *kc = nullptr, *vT = nullptr;     // q [H*d] bf16; K cache [Hkv][seq_alloc][d]; V^T cache [Hkv][d][seq_alloc]
    const StepState *st = nullptr;
    const void *Wo = nullptr;                                  // [N, K = H*d] bf16
    float *out = nullptr;
    int H = 0, Hkv = 0, seq_alloc = 0, N = 0, K = 0;
    float scale = 0.f;
    const LLTable *ll = nullptr; int ll_slot = 0;
};
bool attn_oproj_rep_supported(int64_t H, int64_t Hkv, int64_t d, int64_t h, int64_t max_seq, bool any_size = false);
int launch_attn_oproj_rep(Launcher &L, const AttnRepArgs &a);

// ---- weight conversion at model build ---------------------------------------------------------
// dst[row_map(r)][c] = cvt(src[r0+r][c0+c]); row_mode 0: dst_row0+r, 1: gate rows, 2: up rows
int launch_convert_slice(Launcher &L, int src_dtype, const void *src, int64_t src_ld, int64_t r0, int64_t c0,
                         int64_t rows, int64_t cols, int dst_dtype, void *dst, int64_t dst_ld,
                         int64_t dst_row0, int row_mode, int head_pad = 0, int64_t dm = 1, int64_t d = 1);   // head_pad: padded head_dim (model.hip)
int launch_convert_vec_f32(Launcher &L, int src_dtype, const void *src, int64_t off, int64_t n, float *dst);

}  // namespace fl
