/*
 * oracle/ref_forward.h -- CPU restatement of the FastLLM decoder forward pass.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under fastllm_amd/ (the product) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and there only as the checker.
 *
 * PARITY STATUS: "parity unpinned" at the candle boundary.  The reference
 * (lukehinds/fastllm) is a Rust crate whose arithmetic lives in the un-vendored
 * third-party crates candle-core / candle-nn / candle-transformers ^0.8.2
 * (/root/reference/Cargo.toml:19-21; Cargo.lock is git-ignored, .gitignore:6).
 * No Rust toolchain exists in the build image, and none of the reference's 31
 * unit tests calls forward() or holds a golden logit.
 * This file therefore restates the published Llama / Mistral / Qwen2
 * architecture with candle 0.8.x's numeric choices and
 * is anchored on the reference's own call sites:
 *   src/models/llama.rs:31-50,98-149     (config map, Cache::new, forward(pos))
 *   src/models/mistral.rs:93-154,206-236 (config map, call-counter offset)
 *   src/models/qwen.rs:30-56,123-151     (config map, call-counter offset)
 *   src/models/config.rs:6-54            (BaseModelConfig + validation)
 *   src/models/mod.rs:268-340,363-463    (generate loop)
 * It is cross-checked against an independent public implementation of the same
 * architectures (HuggingFace transformers, fp32 eager) through the committed
 * fixtures in tests/golden/ (generator: tests/golden/make_golden.py).
 */
#ifndef ORACLE_REF_FORWARD_H
#define ORACLE_REF_FORWARD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_LLAMA = 0, ORC_MISTRAL = 1, ORC_QWEN2 = 2 };
enum { ORC_F32 = 0, ORC_BF16 = 1, ORC_F16 = 2 };

/* Fields of the reference's per-family ConfigFile / BaseModelConfig
 * (llama.rs:18-29, mistral.rs:80-92, config.rs:6-18).  0 means "absent in
 * config.json" and takes the reference's default. */
typedef struct orc_config {
    int32_t family;
    int32_t qkv_bias;              /* Qwen2: 1 */
    int64_t hidden_size;
    int64_t intermediate_size;
    int64_t vocab_size;
    int64_t num_hidden_layers;
    int64_t num_attention_heads;
    int64_t num_key_value_heads;   /* 0 -> num_attention_heads */
    int64_t head_dim;              /* 0 -> hidden/heads; explicit for TP shards */
    int64_t max_position_embeddings;
    int64_t sliding_window;        /* 0 -> family default (4096 mistral/qwen, none llama); <0 none */
    double  rms_norm_eps;
    double  rope_theta;            /* 0 -> 10000 */
} orc_config;

typedef struct orc_tensor {
    const char *name;              /* HF name, e.g. model.layers.0.self_attn.q_proj.weight */
    int32_t dtype;                 /* ORC_F32 / ORC_BF16 / ORC_F16 */
    int32_t ndim;
    int64_t shape[4];
    const void *data;              /* host, borrowed for the call */
} orc_tensor;

typedef struct orc_model orc_model;
typedef struct orc_cache orc_cache;

/* all-reduce(sum) hook for tensor-parallel shard runs (tests only); NULL = none */
typedef void (*orc_allreduce_fn)(float *buf, size_t n, void *ctx);

/* round_bf16 = 0: pure fp32 arithmetic (the "reference CPU provider" mode).
 * round_bf16 = 1: fp32 arithmetic, but activations are rounded to bf16 at the
 *   tensor boundaries where the MI355X bf16 path stores bf16 (x*norm_weight, q/k/v
 *   after RoPE, attention output, SiLU-gate product).  Residual stream, norms,
 *   softmax and logits stay fp32.
 * round_bf16 = 2: candle's bf16 execution (the reference hard-wires DType::BF16, main.rs:120): every op's output is a
 *   bf16 tensor -- residual stream, norm (m rounded, divide and multiply each rounded), every Linear, RoPE tables
 *   (Mistral / Qwen2: inv_freq AND positions cast to bf16 before the outer product), Mistral / Qwen2 scores, scaling and
 *   probabilities (Llama runs attention in f32), SiLU, products, logits ([UPSTREAM-RECALLED]:
 *   the candle crates are not under /root/reference).  Measures the distance between the product's bf16 logits and the
 *   reference's bf16 run; never a pass/fail oracle on its own. */
int  orc_model_create(const orc_config *cfg, const orc_tensor *tensors, size_t n,
                      int round_bf16, orc_model **out);
void orc_model_destroy(orc_model *m);
void orc_model_set_allreduce(orc_model *m, orc_allreduce_fn fn, void *ctx);
void orc_model_set_threads(orc_model *m, int nthreads);
int  orc_model_threads(const orc_model *m);

int    orc_cache_create(const orc_model *m, size_t max_seq, orc_cache **out);
void   orc_cache_reset(orc_cache *c);
size_t orc_cache_len(const orc_cache *c);
void   orc_cache_destroy(orc_cache *c);

/* forward(input[1,T], pos, cache) -> last-position logits [V] fp32.
 * `pos` is the RoPE offset of ids[0]; keys/values are appended at cache len
 * (candle's Tensor::cat), so the two may differ (reference quirk C.1). */
int orc_forward(orc_model *m, orc_cache *c, const uint32_t *ids, size_t T, size_t pos,
                float *logits_out);

/* LogitsProcessor ArgMax: iter().enumerate().max_by(total_cmp) -> last max wins */
uint32_t orc_argmax(const float *logits, size_t n);

/* Model<M>::generate (mod.rs:363-463) with temperature 0.
 * pos_mode 0 = "tokens" (pos handed to forward is the token position),
 * pos_mode 1 = "reference": Mistral/Qwen ignore pos and use a per-call counter
 *              (mistral.rs:206-236, qwen.rs:123-151); Llama uses pos (llama.rs:147).
 * eos < 0 disables the EOS check.  step_logits (optional) receives the logits
 * each sampled token was drawn from, [n_out][V].  Returns tokens emitted. */
int orc_generate(orc_model *m, orc_cache *c, const uint32_t *prompt, size_t T,
                 size_t max_tokens, int64_t eos, int pos_mode,
                 uint32_t *out_tokens, float *step_logits);

const char *orc_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
