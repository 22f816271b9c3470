/*
 * fastllm_mi355x.h -- C ABI of the MI355X (gfx950 / CDNA4) decoder forward-pass backend
 * for FastLLM (lukehinds/fastllm).
 *
 * This is the drop-in boundary for ONE hot path of the reference: the batch-1, KV-cached
 * causal-LM forward pass behind
 *     trait ModelInitializer { initialize_model; initialize_cache; forward }
 *     (/root/reference/src/models/model_initializer.rs:6-22)
 * for the three families the reference registers (LlamaWithConfig llama.rs:94-150,
 * MistralWithConfig mistral.rs:156-237, QwenWithConfig qwen.rs:89-152).  The reference has no
 * FFI today (no extern "C" anywhere); its arithmetic is delegated to candle.  A Rust
 * `impl ModelInitializer for Mi355xWithConfig` binds exactly the entry points below
 * (binding shown in INTEGRATION.md).
 *
 * Conventions
 *   - every entry returns an fl_status (0 = OK, negative = error); fl_last_error() gives a
 *     thread-local message (anyhow::Error analogue).  Nothing aborts or throws across the ABI: every
 *     entry is an exception barrier (std::bad_alloc -> FL_ERR_OOM, anything else -> FL_ERR_HIP).
 *   - plain pointers and sizes only; opaque handles for model and cache.
 *   - thread-safe: any number of threads may call fl_forward on ONE model with DISTINCT caches
 *     (the reference's streaming path does that, mod.rs:137-238); submission is serialised
 *     per model.  Every entry sets the HIP device itself.
 *   - the library never falls back to a CPU path: without a usable gfx950 device
 *     fl_model_create fails with FL_ERR_NO_DEVICE.
 */
#ifndef FASTLLM_MI355X_H
#define FASTLLM_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FL_ABI_VERSION 2

typedef enum fl_status {
    FL_OK = 0,
    FL_ERR_BAD_CONFIG = -1,      /* reference: assert!/expect panics, mistral.rs:109-127, qwen.rs:32-37 */
    FL_ERR_MISSING_TENSOR = -2,  /* candle VarBuilder "cannot find tensor" */
    FL_ERR_SHAPE_MISMATCH = -3,
    FL_ERR_OOM = -4,
    FL_ERR_HIP = -5,
    FL_ERR_RCCL = -6,
    FL_ERR_SEQ_OVERFLOW = -7,
    FL_ERR_BAD_ARGUMENT = -8,
    FL_ERR_NO_DEVICE = -9,
    FL_ERR_UNSUPPORTED = -10
} fl_status;

/* ModelArchitecture::get_family (llama.rs:153, mistral.rs:240, qwen.rs:174) */
typedef enum fl_family { FL_FAMILY_LLAMA = 0, FL_FAMILY_MISTRAL = 1, FL_FAMILY_QWEN2 = 2 } fl_family;

/* candle_core::DType subset the reference can hand over (dtype_utils.rs:10-23) */
typedef enum fl_dtype { FL_DTYPE_F32 = 0, FL_DTYPE_BF16 = 1, FL_DTYPE_F16 = 2 } fl_dtype;

/* The fields of the reference's ConfigFile / BaseModelConfig (llama.rs:18-29,
 * mistral.rs:80-92, config.rs:6-18).  A 0 in an optional field means "absent from
 * config.json" and takes the reference's default:
 *   num_key_value_heads  -> num_attention_heads        (llama.rs:39, mistral.rs:97, qwen.rs:45)
 *   rope_theta           -> 10000                      (llama.rs:41, mistral.rs:137, qwen.rs:47)
 *   max_position_embeddings -> 4096 llama / 32768 mistral, qwen (llama.rs:47, mistral.rs:138, qwen.rs:48)
 *   sliding_window       -> 4096 mistral, qwen; unused for llama (mistral.rs:139, qwen.rs:49)
 * head_dim is hidden_size / num_attention_heads (mistral.rs:67-76, config.rs:32-44); it is not
 * a config field in the reference and is validated, not passed.  Every even head_dim up to 128 is supported, as the
 * reference accepts them (config.rs:31-43): the kernels are built for 64 and 128 (TinyLlama; Mistral-7B, Qwen2-7B and every
 * Llama-2/3 size) and any other value -- 48, 80, 96, 100 (OpenLLaMA-3B) ... -- runs as the next of the two, with zero
 * weight rows between the halves of every head (fl_model_info.head_dim still reports the model's).  REFUSED with
 * FL_ERR_UNSUPPORTED at fl_model_create: head_dim above 128, and hidden_size that is not a multiple of 8; an odd
 * head_dim is FL_ERR_BAD_CONFIG, as in the reference. */
typedef struct fl_config {
    int32_t family;                 /* fl_family */
    int32_t qkv_bias;               /* 1 for Qwen2 (q/k/v_proj.bias tensors), else 0 */
    int64_t hidden_size;
    int64_t intermediate_size;
    int64_t vocab_size;
    int64_t num_hidden_layers;
    int64_t num_attention_heads;
    int64_t num_key_value_heads;
    int64_t max_position_embeddings;
    int64_t sliding_window;
    double  rms_norm_eps;
    double  rope_theta;
} fl_config;

/* One entry of initialize_model's HashMap<String, Tensor> (model_initializer.rs:12).
 * Borrowed for the duration of fl_model_create only; the library copies, shards and
 * re-lays-out into HBM.  `device` < 0: `data` is host memory; >= 0: `data` is a device
 * pointer on that HIP device (the reference loads safetensors straight onto `device`,
 * huggingface.rs:88,125); fl_model_create synchronises the devices before it reads such
 * tensors, so work still in flight on the caller's streams is waited for. */
typedef struct fl_tensor {
    const char *name;               /* HF name, e.g. "model.layers.0.self_attn.q_proj.weight" */
    int32_t dtype;                  /* fl_dtype */
    int32_t ndim;
    int64_t shape[4];
    const void *data;
    int32_t device;
    int32_t _pad;
} fl_tensor;

/* Tensor-parallel placement.  New capability (the reference is single-device, README.md:149). */
typedef enum fl_tp_mode {
    FL_TP_NONE = 0,                 /* one GPU: device_ids[0] (or device 0 if NULL) */
    FL_TP_SINGLE_PROCESS = 1,       /* this process drives tp_size GPUs (device_ids[tp_size]); the
                                       shape of the reference's one-process server (main.rs:128).  Decode
                                       collectives are one-shot pushes over peer pointers, one hipGraph per
                                       shard; RCCL group calls carry the large prefill collectives */
    FL_TP_MULTI_PROCESS = 2,        /* one process per GPU: this process is tp_rank of tp_size.  With a
                                       unique_id it joins that RCCL communicator (large prefill
                                       collectives) and connects the peer inboxes for the small decode
                                       collectives by itself; with unique_id == NULL there is no RCCL and
                                       the host connects the inboxes: fl_comm_ipc_export / _connect */
    FL_TP_EMULATED = 3              /* tp_size shards on ONE GPU, collectives done locally: lets a
                                       single-GPU box verify the sharded kernels (tests only) */
} fl_tp_mode;

#define FL_UNIQUE_ID_BYTES 128
typedef struct fl_parallel {
    int32_t mode;                   /* fl_tp_mode */
    int32_t tp_size;
    int32_t tp_rank;                /* FL_TP_MULTI_PROCESS only */
    int32_t n_device_ids;
    const int32_t *device_ids;
    const void *unique_id;          /* FL_TP_MULTI_PROCESS: FL_UNIQUE_ID_BYTES from fl_comm_unique_id */
} fl_parallel;

typedef struct fl_model fl_model;
typedef struct fl_cache fl_cache;

int         fl_abi_version(void);
const char *fl_last_error(void);
int         fl_device_count(int *count);
/* ncclGetUniqueId; rank 0 calls it and ships the bytes to the other ranks out of band */
int         fl_comm_unique_id(void *out /* FL_UNIQUE_ID_BYTES */);

/* ModelInitializer::initialize_model(&Config, HashMap<String,Tensor>, DType, &Device)
 * (model_initializer.rs:10-17; call site huggingface.rs:135).  compute_dtype: FL_DTYPE_BF16 is
 * the reference's hard-wired dtype (main.rs:120); FL_DTYPE_F32 is the fp32 parity mode. */
int fl_model_create(const fl_config *cfg, const fl_tensor *tensors, size_t n_tensors,
                    int32_t compute_dtype, const fl_parallel *par, fl_model **out);

/* FL_TP_MULTI_PROCESS: the decode step's two [h] fp32 all-reduces per layer and the logits all-gather
 * are one-shot pushes into inboxes that every rank maps from every peer's HBM over xGMI (new
 * capability; the reference is single-device, README.md:149).  A model created with a unique_id
 * wires them itself over RCCL.  Otherwise: every rank exports its inbox handle, the host all-gathers
 * the tp handles out of band (rank order) and every rank connects before its first forward.
 * Connected groups with one GPU per rank exchange the all-reduces inside the o_proj / down_proj GEMV epilogues
 * (fl_model_info.fused_all_reduce); waits for a peer are bounded (FL_AR_TIMEOUT_MS) and end in FL_ERR_RCCL. */
#define FL_IPC_HANDLE_BYTES 64
int fl_comm_ipc_export(fl_model *m, void *handle_out /* FL_IPC_HANDLE_BYTES */);
int fl_comm_ipc_connect(fl_model *m, const void *handles /* tp_size * FL_IPC_HANDLE_BYTES */);
/* Clone for the streaming path (mod.rs:155,181,207) is a refcount bump. */
void fl_model_retain(fl_model *m);
void fl_model_release(fl_model *m);

typedef struct fl_model_info {
    fl_config cfg;                  /* defaults resolved */
    int64_t head_dim;
    int32_t compute_dtype;
    int32_t tp_size;
    int64_t weight_bytes_per_token; /* algorithmic HBM bytes a decode step reads from weights (whole model) */
    int64_t kv_bytes_per_position;  /* K+V bytes one cached position adds to a decode step */
    int64_t hbm_bytes_allocated;    /* this process, all shards */
    int32_t small_collectives;      /* decode collectives: 0 none (tp 1), 1 RCCL, 2 one-shot peer inboxes, 3 local (emulated) */
    int32_t fused_all_reduce;       /* 1: decode all-reduces ride in the o_proj / down_proj GEMV epilogues (no kernel of their own) */
    int32_t rccl_ranks;             /* ncclCommCount of this rank's RCCL communicator; 0: no communicator (tp 1, emulated, IPC-only groups) */
    int32_t _reserved;
} fl_model_info;
int fl_model_get_info(const fl_model *m, fl_model_info *out);

/* ModelInitializer::initialize_cache(&Device, DType) (model_initializer.rs:19; per request,
 * mod.rs:370).  The reference's version cannot see the model (hard-coded TinyLlama dims,
 * llama.rs:125-145); here the cache is derived from the model and owned by the caller. */
int    fl_cache_create(fl_model *m, size_t max_seq, fl_cache **out);
void   fl_cache_reset(fl_cache *c);          /* clear_kv_cache (mistral.rs:220, qwen.rs:148) */
size_t fl_cache_len(const fl_cache *c);
size_t fl_cache_capacity(const fl_cache *c);
void   fl_cache_destroy(fl_cache *c);

/* ModelInitializer::forward(&self, input[1,T] u32, pos, &mut cache) -> logits
 * (model_initializer.rs:21; call sites mod.rs:402-405,446-451).
 *   ids[T]      token ids (batch is always 1, mod.rs:283-291)
 *   pos         RoPE offset of ids[0].  Keys/values are APPENDED at fl_cache_len (candle's
 *               Tensor::cat), so a caller that passes the reference's call counter for
 *               Mistral/Qwen (mistral.rs:226,234) gets the reference's results.
 *   logits_out  host, [vocab_size] fp32: the LAST position's logits, i.e. what the caller takes
 *               with logits.get(0)?.flatten_all()? (mod.rs:305,421) after to_dtype(F32).
 * Blocking: returns when logits are on the host. */
int fl_forward(fl_model *m, fl_cache *c, const uint32_t *ids, size_t T, size_t pos, float *logits_out);

/* Same forward, but LogitsProcessor ArgMax (temperature < 1e-7; ties -> LAST maximal index,
 * Rust Iterator::max_by) is evaluated on the device and only the token id comes back. */
int fl_forward_argmax(fl_model *m, fl_cache *c, const uint32_t *ids, size_t T, size_t pos, uint32_t *token_out);

/* The body of the greedy loop of Model<M>::generate (mod.rs:411-453) kept on the device:
 * starting from `first_token` (already sampled by the caller from the prefill logits), run
 * n_steps x { forward([tok], pos) ; tok = argmax ; pos += pos_stride } with no host round trip.
 * tokens_out[i] is the token sampled after step i.  Stops early (n_out < n_steps) when the
 * sampled token == eos (eos < 0: never).  pos_stride is 1; pos is whatever offset the caller's
 * position mode dictates. */
int fl_decode_greedy(fl_model *m, fl_cache *c, uint32_t first_token, size_t pos, size_t n_steps,
                     int64_t eos, uint32_t *tokens_out, size_t *n_out);

/* LogitsProcessor::new(seed, Some(temperature), None) + .sample() (mod.rs:373-374,425-428) on the device.
 * temperature < 1e-7 is ArgMax, as in candle.  Otherwise Sampling::All: softmax(logits / temperature) in
 * fp32, then rand 0.8's WeightedIndex<f32> driven by StdRng::seed_from_u64(seed) (ChaCha12); one u32 word of
 * the stream is consumed per sampled token.  The reference builds a fresh processor with seed 0 per
 * request, samples the first token from the prefill logits and goes on in the loop: here that is
 *   fl_forward_sample(.., {t, 0, 0}, &tok);  fl_decode_sample(.., tok, pos, n, eos, {t, 0, 1}, ..)
 * (draws_done = words already consumed, so one request may span several calls). */
typedef struct fl_sampling {
    double   temperature;
    uint64_t seed;
    uint64_t draws_done;
} fl_sampling;
int fl_forward_sample(fl_model *m, fl_cache *c, const uint32_t *ids, size_t T, size_t pos,
                      const fl_sampling *sampling, uint32_t *token_out);
int fl_decode_sample(fl_model *m, fl_cache *c, uint32_t first_token, size_t pos, size_t n_steps, int64_t eos,
                     const fl_sampling *sampling, uint32_t *tokens_out, size_t *n_out);

/* Batched decode: B <= 64 caches of one model advanced together, one read of the weights per step for all
 * of them.  New capability: the reference runs concurrent streams as independent single-sequence loops
 * (mod.rs:137-238), each paying for the whole weight stream.  Every sequence keeps its own cache, RoPE
 * position and sampler state; per sequence the results are those of the single-sequence entry points
 * (same kernels' arithmetic, up to fp32 summation order in the norm).  bf16 or fp32 (fp32, and bf16 caches outside the MFMA attention
 * layout: RoPE and attention run per sequence, the projections once for all rows); one GPU, or -- bf16 -- the ranks of an FL_TP_MULTI_PROCESS group
 * with connected inboxes -- every rank then builds the same batch over its own caches and calls the same entry points in the same
 * order (the step's all-reduces and the gather of the ranks' logits blocks are collectives; a rank that stays away is
 * FL_ERR_RCCL after FL_AR_TIMEOUT_MS on the others).
 * The caches stay usable on their own (prefill them with fl_forward*, then batch the decode). */
typedef struct fl_batch fl_batch;
int  fl_batch_create(fl_model *m, fl_cache *const *caches, size_t n, fl_batch **out);
void fl_batch_destroy(fl_batch *b);
/* Continuous batching: `cache` (prefilled by fl_forward*, of the same model, not in the batch yet) takes the place of sequence
 * `slot` -- a stream that hit its EOS leaves, a waiting one joins -- without rebuilding the batch: one small copy, the step's captured
 * graph stays valid.  The cache that leaves is untouched and stays usable on its own.  A slot with no stream to serve can hold a
 * spare cache (its row is computed and ignored).  Not on a tensor-parallel group's batch while a step is outstanding on a peer:
 * every rank replaces the same slot between the same two steps.  (mod.rs:137-238: streams come and go independently.) */
int  fl_batch_replace(fl_batch *b, size_t slot, fl_cache *cache);
/* one step: tokens[i] at RoPE offset pos[i]; logits_out [n][V] fp32 host or NULL; argmax_out [n] or NULL */
int  fl_batch_forward(fl_batch *b, const uint32_t *tokens, const size_t *pos, float *logits_out, uint32_t *argmax_out);
/* the loop of fl_decode_greedy / fl_decode_sample for every sequence: tokens_out [n][n_steps], n_out [n];
 * a sequence stops counting at its EOS (its cache length is that of fl_decode_greedy); sampling may be NULL
 * (ArgMax); with sampling every sequence draws from its own copy of the seeded stream, as every request of
 * the reference does (seed 0 per request, mod.rs:373-374) */
int  fl_batch_decode(fl_batch *b, const uint32_t *first_tokens, const size_t *pos, size_t n_steps, int64_t eos,
                     const fl_sampling *sampling, uint32_t *tokens_out, size_t *n_out);
/* ... with every sequence's own EOS id (eos[i] < 0: none; eos NULL: none for all) and its own sampler (sampling[i]: temperature < 1e-7
 * is ArgMax; sampling NULL: ArgMax for all) -- a request's temperature is its own (chat.rs:24-25), and one batch serves requests that
 * differ in it.  draws_done lets a request span several calls, as in fl_decode_sample. */
int  fl_batch_decode_each(fl_batch *b, const uint32_t *first_tokens, const size_t *pos, size_t n_steps, const int64_t *eos,
                          const fl_sampling *sampling, uint32_t *tokens_out, size_t *n_out);

int fl_synchronize(fl_model *m);

/* Which slice of a full HF tensor does tp_rank own?  Pure host function (no GPU):
 * out = {row_begin, row_end, col_begin, col_end}.  Column-parallel q/k/v/gate/up/lm_head
 * (rows of the [out,in] matrix), row-parallel o_proj/down_proj (columns), everything else whole. */
int fl_tp_slice(const fl_config *cfg, const char *tensor_name, int32_t tp_rank, int32_t tp_size,
                int64_t out[4]);

/* ---- measurement hooks (bench.py / profiles) --------------------------------------------- */
typedef struct fl_kernel_stat {
    char    name[48];               /* kernel class, e.g. "gemv_bf16" */
    int64_t launches;
    double  total_ms;               /* sum of HIP-event durations (hipExtLaunchKernelGGL start/stop) */
    double  bytes;                  /* algorithmic HBM bytes over those launches */
    double  flops;                  /* algorithmic flops over those launches */
} fl_kernel_stat;
/* While profiling is on, forwards run eagerly and every kernel launch is bracketed by a HIP
 * event pair on the launch stream. */
int fl_profile_begin(fl_model *m);
int fl_profile_end(fl_model *m, fl_kernel_stat *stats, size_t cap, size_t *n_stats);

/* Times the small all-reduce of a decode step (n fp32 values, n <= hidden_size) on the links the group really has, one form at a
 * time: form 0 = ncclAllReduce (RCCL), 1 = the one-shot kernel over peer-mapped HBM (k_comm.hip), 2 = the exchange that rides in
 * the GEMV epilogues (comm_ll.h; timed through its stand-alone exerciser, one-shot epoch steps subtracted).  Collective: every
 * rank of the group calls it with the same arguments.  *us_per_call < 0: that form is not available in this group.
 * (the decode all-reduce after o_proj / down_proj; bench.py --gpus N reports the three side by side.) */
int fl_comm_probe(fl_model *m, int32_t form, int64_t n, int32_t iters, double *us_per_call);

/* Proves the one-shot collectives of a connected FL_TP_MULTI_PROCESS group on data whose sums are exact: one all-reduce of n
 * integer-valued floats (n <= FL_AR_INBOX_FLOATS; 16 384 and more take the many-workgroup form, fewer the one-workgroup form),
 * every wait bounded by 2 s.  Collective: every rank calls it with the same n.  *ok = 1: this rank holds the exact sums; 0: a wrong
 * sum or a wait that gave up (the error state is cleared: the group stays usable).  A model created with a unique_id has run this
 * itself (and fallen back by an all-ranks vote); a group wired by fl_comm_ipc_connect can ask for it here.
 * (the all-reduce after o_proj / down_proj.) */
int fl_comm_selftest(fl_model *m, int64_t n, int32_t *ok);

/* Process-wide switches (sweeps and tests; not needed in normal use).  Every switch is an integer row of ONE table
 * (fastllm_amd/csrc/common.h, enum TuneKey, documents each; DESIGN.md's appendix lists them): key "gemm_h4" is the row read from
 * the environment variable FL_GEMM_H4.  The environment is read ONCE, on first use; afterwards only this call changes a switch.
 *   value        >= 0, or -1 = "automatic" for the switches that have such a setting
 *   "reload_env" re-reads every FL_<NAME> from the environment (value ignored)
 *   "gemv_blocks" / "gemv_waves"  force the decode GEMV's grid / waves per workgroup (0 = automatic); "gemv_r" rows per wave pass
 *                (2|4) and "gemv_u" 512-element chunks per pipeline block (0 automatic, 2|4|7|8) are table rows like the rest
 *   "experimental"  FL_OK in the EXPERIMENTAL build (make EXPERIMENTAL=1), FL_ERR_UNSUPPORTED in the default one
 * FL_ERR_BAD_ARGUMENT: unknown key.  FL_ERR_UNSUPPORTED: the key belongs to a kernel that is compiled into the EXPERIMENTAL build
 * only (decode engine, fused attention + o_proj, attention prefetch workgroups, loader waves, "engine_grid"); the default build's
 * environment cannot reach those either. */
int fl_tune(const char *key, int value);

/* y[T,N] = x[T,K] . W[N,K]^T (+bias): the projection kernel family on host buffers, for unit
 * tests and micro-benchmarks.  dtype is the storage type of x and W (bf16 or f32); y is fp32.
 * epilogue: 0 none, 1 silu-gate (W rows are gate/up pairs in HF order: gate = rows [0,N/2),
 * up = rows [N/2,N); y is [T, N/2]).  iters > 0 with ms_out != NULL times `iters` launches. */
int fl_op_linear(const void *x, const void *w, const float *bias, int64_t T, int64_t N, int64_t K,
                 int32_t dtype, int32_t epilogue, float *y, int32_t iters, double *ms_out);

/* The token-selection kernel alone, for unit tests: `n_draws` successive selections from one host logits
 * vector (consuming successive words of the seeded stream; ArgMax when temperature < 1e-7). */
int fl_op_sample(const float *logits, int64_t V, const fl_sampling *sampling, int64_t n_draws, uint32_t *tokens_out);

/* The bf16 MFMA attention kernels alone, for unit tests against an fp64 reference (they are otherwise only seen through
 * whole-model logits).  One sequence: q [T][H*d] (RoPE already applied), k / v [s_past + T][Hkv*d], all bf16 row-major; the
 * first s_past positions are the cache, the last T the new tokens.  Mask as the forward pass applies it:
 * T == 1: no mask; T > 1: cached keys visible, new key j visible to query t iff j <= t and j + window >= t (window < 0: no
 * window).  kernel: 0 = what the model would launch, 1 = decode kernel (T must be 1), 2 = 16-row prefill kernel, 3 =
 * 32-row prefill kernel.  nsplit: key splits of the decode kernel (0 = as fl_cache_create picks; 1 = one wide workgroup
 * per kv head).  out [T][H*d] fp32 (the kernels' bf16 output widened). */
int fl_op_attention(const void *q, const void *k, const void *v, int64_t T, int64_t s_past, int64_t H, int64_t Hkv,
                    int64_t d, int64_t window, int32_t kernel, int32_t nsplit, float *out);

#ifdef __cplusplus
}
#endif
#endif
