// common.h -- shared host/device helpers for the MI355X (gfx950) forward-pass library.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fastllm_mi355x.h"

namespace fl {

// ---- errors -------------------------------------------------------------------------------
void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
const char *last_error();

#define FL_FAIL(code, ...) do { ::fl::set_error(__VA_ARGS__); return (code); } while (0)
#define FL_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    ::fl::set_error("HIP error %s at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #expr); \
    return (e_ == hipErrorOutOfMemory) ? FL_ERR_OOM : FL_ERR_HIP; } } while (0)
#define FL_TRY(expr) do { int rc_ = (expr); if (rc_ != FL_OK) return rc_; } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to (function, device): raise it for `fn` on the CURRENT device when
// a launch needs more than the 64 KiB default (remembered per pair; a single-process tensor-parallel group launches
// the same instantiation on every GPU of the group).
int raise_dynamic_lds(const void *fn, size_t lds_bytes);

// ---- process-wide switches -------------------------------------------------------------------
// Every kernel-selection knob is an int in ONE table (model.hip, g_tune_table): read once from the environment (FL_<NAME>) when first
// used, afterwards changed only through fl_tune(name, value) ("reload_env" re-reads the environment: tests and A/B tools).
// No per-launch getenv.
enum TuneKey {
    TK_GEMM_H4,   // 128 x 256 GEMM with in-launch K-slice sums: 0 never, 1 where the rule prefers it, 2 wherever it is supported
    TK_GEMM_W14,   // 256 x 224 four-wave GEMM (k_gemm_w14.hip): 0 never, 1 where a 224-column grid fills the chip and the 256-column one does not, 2 wherever supported
    TK_GEMM_ROPE_4W,   // long prompts (>= 768 tokens): the QKV projection's RoPE / bias / KV append in the four-wave kernel's epilogue (0: fp32 output + rope_kv_append)
    TK_GEMM_F32_MFMA,   // fp32 mode: prompt GEMMs on v_mfma_f32_32x32x2_f32 (k_gemm_f32.hip); 0: the 64 x 64 VALU kernel
    TK_W14_NT,   // 256 x 224 kernel: W pieces non-temporal (1), plain (0), -1: up to 512 tokens (two row tiles per W panel)
    TK_H4_NT,   // 128 x 256 kernel: W pieces non-temporal (1), plain (0), -1: at 176-256 tokens (a W panel read by at most two row tiles; also the 256 x 128 / 256 x 256 kernels' single row tile)
    TK_H4_SPLIT,   // 0: as the rule; 1..4: K slices it uses (probes, tests)
    TK_H4_PF,   // its L2 prefetch of the W panel: K tiles ahead + 256 x lanes per 128-byte line (0 lanes: off; measured neutral)
    TK_H4_WAIT_US,   // how long an early K slice waits for the others before it leaves its blocks to the last one
    TK_OP_MAXSPLIT,   // fl_op_linear: most K slabs (0: as the model)
    TK_OP_LINEAR_DMA,   // fl_op_linear: T <= 8 through the batched-decode ring kernel
    TK_OP_HOT,   // fl_op_linear timing: copies of W the loop rotates over (0: enough to defeat the Infinity Cache)
    TK_AR_INBOX_FLOATS,   // one-shot collectives: floats per inbox slot
    TK_AR_TIMEOUT_MS,   // bound on every wait for a peer
    TK_VERBOSE,   // bootstrap log lines
    TK_TP_FUSED_AR,   // decode all-reduces inside the GEMV epilogues (2: even when ranks share a GPU)
    TK_ATTN_NW,   // VALU decode attention: waves
    TK_ATTN_PREFETCH,   // decode attention: workgroups that touch the next launch's weights (experimental build)
    TK_ATTN_PREFETCH_LINES,
    TK_ATTN_PREFETCH_PCT,
    TK_ATTN_PREFETCH_DELAY,
    TK_ATTN_BATCH_WGS,   // batched decode attention: workgroup cap
    TK_ATTN_PF32_MIN_T,   // 32-row prefill attention from this many tokens (0: 640 with wave pairs, else 1024)
    TK_ATTN_PF32_KS2,   // its key-split form: -1 automatic, 0 / 1
    TK_ATTN_PF32_PAIRED,   // its work distribution: -1 automatic, 0 plain, 1 paired, 2 snake
    TK_ATTN_PF_WAVES,   // 16-row prefill attention: waves per workgroup (0: automatic)
    TK_ATTN_PF_STAGES,
    TK_ATTN_PF_KSPLIT,
    TK_AO_DELAY,   // fused attention + o_proj (experimental build): head start of the attention loads, x 1/4 us
    TK_AO_WAVES,
    TK_ENGINE_DELAY,   // decode engine (experimental build)
    TK_ENGINE_PF,
    TK_ENGINE_TIMEOUT_MS,
    TK_SK_MINSTEPS,   // stream-K: K steps per piece, at least
    TK_GEMM_4W,   // four-wave 256 x 256 GEMM: 0 never, 1 from 768 tokens on wide matrices, 2 always
    TK_GEMM_GROUPM,   // its tile order: row tiles per group (0: 4)
    TK_8P_MINK,   // 256 x 256 GEMM: K steps per slice, at least
    TK_GEMM_8P,   // 256 x 256 GEMM: 0 off, 1 where the cost model prefers it, 2 always
    TK_GEMM_256,
    TK_GEMM_256_SPLIT,
    TK_GEMM_STREAMK,   // 1: peeled tails; 0: tails on 128 x 128 tiles; 2: also whole grids of 96-255 tiles; 3: every shape (tests)
    TK_GEMM_PEEL,
    TK_GEMM_RESID,   // residual epilogue of o_proj / down_proj (0: keep the rmsnorm_add launches)
    TK_GEMM_SKINNY_MAXT,
    TK_SKINNY_STAGES,
    TK_SKINNY_NT,
    TK_SKINNY_WM,
    TK_SKINNY_LOADERS,   // short-prompt GEMM staged by loader waves (experimental build): -1 = only with FL_SKINNY_STAMPS
    TK_GEMM_SKINNY_MAXT2,
    TK_GEMV_SMALL,
    TK_GEMV_R,
    TK_GEMV_U,
    TK_BATCH_U,
    TK_BATCH_MODE,
    TK_BATCH_MFMA_MIN,
    TK_DMA_KT,
    TK_FORCE_GENERIC_GEMM,
    TK_GEMM_SKINNY,
    TK_ROPE_VEC,
    TK_WEIGHT_ARENA,
    TK_KSPLIT_MID,   // most slabs for prompts of 129-1024 tokens (0: eight)
    TK_PREFILL_CHUNK,
    TK_GRAPH,   // decode step as a hipGraph: -1 = on (single GPU), off for an RCCL group of one unless asked
    TK_FUSED,
    TK_ALLOW_ANY_ARCH,
    TK_ENGINE,   // persistent decode engine (experimental build)
    TK_FUSE_OPROJ,   // attention + o_proj in one launch (experimental build)
    TK_ONESHOT,
    TK_DEBUG_RCCL_SELF,
    TK_ATTN_MFMA,
    TK_ATTN_NSPLIT,   // decode attention splits (-1: by cache length)
    TK_ATTN_REP,
    TK_SAMPLE_WALK,
    TK_ARGMAX_FUSED,
    TK_TP_OVERLAP,
    TK_TP_OVERLAP_MIN_T,
    TK_QKV_SPLIT,
    TK_TP_GRAPH,
    TK_BATCH_DMA_MIN,
    TK_H4_OPROJ_1K,   // o_proj of 641-1024-token prompts on the 128 x 256 kernel
    TK_H4_TAIL,   // a peeled GEMM's tail columns on the 128 x 256 kernel instead of stream-K + fix-up: 0 never, 1 with 2-4 in-launch slices, 2 (default) also unsliced when the tail alone fills the chip
    TK_RS_LAZY,   // 1/rms behind a residual epilogue: taken from the partial sums by the consuming projection (0: rms_finalize launch)
    TK_BATCH_UNFUSED_MIN,   // first batch size on the prefill-shaped step (-1: 3 with the ring kernel, else 7)
    TK_DEBUG_RS_PARTS,   // tests: gemm_takes_rs_parts() answers yes for every bf16 prompt shape, so that kernels which cannot take partial sums meet them (rs_parts_to_vector)
    TK_DEBUG_TP_LOOPBACK,   // tools (EXPERIMENTAL build only): an FL_TP_MULTI_PROCESS model without unique_id connects every inbox entry to ITSELF and plays all ranks' pushes (one rank's step, timed with its exchange in place; results meaningless)
    TK_DEBUG_POISON,   // tests: every device allocation of the model / cache / batch objects is filled with this byte before use (255: bf16 / fp32 NaN patterns; 63: finite 0.75s), so that a read of bytes nobody wrote shows at once instead of depending on what the allocator handed back
    TK_GEMM_SKF,   // short prompts / decode batches (2-128 rows) on the kernel whose K slices meet inside the launch (k_gemm_skf.hip): 0 off; 1 (default) a tensor-parallel rank's complete outputs up to 64 rows; 2 (EXPERIMENTAL build only; the default build treats it as 1) also the five-launch layer (residual + norm and RoPE + KV-append epilogues: measured SLOWER than slabs + rmsnorm_add / rope_kv launches, profiles/r05/README.md); 3 also every plain launch_linear shape (tests)
    TK_SKF_SPLIT,   // its K slices: 0 by rule, 1..4 forced (tests)
    TK_PREFILL_DMA,   // prompts of 2-32 tokens: the wide gate/up projection on the LDS-DMA ring kernel of the decode batches (k_gemv_dma.hip); 0: the short-prompt GEMM
    TK_ONESHOT_WIDE,   // one-shot collectives of 16384 floats and more (a decode batch's deltas and logits, a short prompt's span) over up to 256 workgroups (N > 1: at most N); 0: the one-workgroup kernel
    TK_F32_ROWS_MAX,   // fp32 mode: projections of 2 ... this many token rows (short prompts, decode batches; at most 64) as a weight stream with the x rows in LDS (gemv_f32_rows_kernel); 0: the MFMA tiles
    TK_GATEUP_ROWSPLIT,   // gate/up of a prompt 1-96 tokens past an EVEN number of 256-row tiles (513-608, 1025-1120 ...): whole rounds on the 224-column kernel + the last rows as a launch of their own; 0: one launch
    TK_COUNT
};
int tune(TuneKey k);
const char *env_str(const char *name);   // diagnostic file paths (FL_*_STAMPS) and the fault injector: null when unset or empty
int tune_set(const char *name, int value);   // FL_OK, or FL_ERR_BAD_ARGUMENT for an unknown name
void tune_reload_env();

// ---- element types ------------------------------------------------------------------------
typedef uint16_t bf16_t;   // raw bf16 bits in memory

__host__ __device__ inline float bf16_bits_to_float(bf16_t b) {
    union { uint32_t u; float f; } v; v.u = (uint32_t)b << 16; return v.f;
}
__device__ inline bf16_t float_to_bf16_bits(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 on gfx950 (RNE, NaN stays NaN)
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
__host__ inline bf16_t float_to_bf16_bits_host(float f) {
    union { uint32_t u; float f; } v; v.f = f;
    if ((v.u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((v.u >> 16) | 0x40);
    return (bf16_t)((v.u + 0x7fffu + ((v.u >> 16) & 1u)) >> 16);
}

template <typename T> struct elem;
template <> struct elem<float> {
    static constexpr int kDtype = FL_DTYPE_F32;
    __device__ static float ld(const float *p) { return *p; }
    __device__ static void st(float *p, float v) { *p = v; }
};
template <> struct elem<bf16_t> {
    static constexpr int kDtype = FL_DTYPE_BF16;
    __device__ static float ld(const bf16_t *p) { return bf16_bits_to_float(*p); }
    __device__ static void st(bf16_t *p, float v) { *p = float_to_bf16_bits(v); }
};

// 8 consecutive elements as floats; p must be 16-byte aligned (bf16) / 32-byte (f32: two 16-B loads)
typedef float float8 __attribute__((ext_vector_type(8)));
typedef float float4v __attribute__((ext_vector_type(4)));
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));

__device__ inline void unpack8(const uint4v r, float (&o)[8]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        o[2 * i] = __uint_as_float(r[i] << 16);
        o[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
    }
}
__device__ inline void load8(const bf16_t *p, float (&o)[8]) {
    uint4v r = *reinterpret_cast<const uint4v *>(p);
    unpack8(r, o);
}
__device__ inline void load8(const float *p, float (&o)[8]) {
    float4v a = *reinterpret_cast<const float4v *>(p);
    float4v b = *reinterpret_cast<const float4v *>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; i++) { o[i] = a[i]; o[4 + i] = b[i]; }
}
// streaming (read-once) variants for weights: non-temporal so they do not displace L2/MALL lines
__device__ inline void load8_nt(const bf16_t *p, float (&o)[8]) {
    uint4v r = __builtin_nontemporal_load(reinterpret_cast<const uint4v *>(p));
    unpack8(r, o);
}
__device__ inline void load8_nt(const float *p, float (&o)[8]) {
    float4v a = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(p));
    float4v b = __builtin_nontemporal_load(reinterpret_cast<const float4v *>(p + 4));
#pragma unroll
    for (int i = 0; i < 4; i++) { o[i] = a[i]; o[4 + i] = b[i]; }
}
__device__ inline void store8(bf16_t *p, const float (&v)[8]) {
    uint4v r;
#pragma unroll
    for (int i = 0; i < 4; i++)
        r[i] = (uint32_t)float_to_bf16_bits(v[2 * i]) | ((uint32_t)float_to_bf16_bits(v[2 * i + 1]) << 16);
    *reinterpret_cast<uint4v *>(p) = r;
}
__device__ inline void store8(float *p, const float (&v)[8]) {
    float4v a, b;
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = v[i]; b[i] = v[4 + i]; }
    *reinterpret_cast<float4v *>(p) = a;
    *reinterpret_cast<float4v *>(p + 4) = b;
}

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Per-request step state kept on the device so a captured decode graph can be replayed with
// no argument update: kernels read token / RoPE offset / KV length from here.
struct StepState {
    uint32_t token;      // id fed to the next forward (decode)
    uint32_t pos;        // RoPE offset of the first token of this call
    uint32_t len;        // KV entries already cached before this call
    uint32_t step;       // decode steps done in the current fl_decode_greedy call
    int32_t  eos;        // -1: none
    uint32_t done;       // set once eos was sampled
    uint32_t error;      // device-side failure code (bounded spin gave up): the host turns it into FL_ERR_HIP
    uint32_t call0;      // KV entries cached when the API call began (== len unless the library chunks a prefill)
};

// Token selection state of a cache, next to StepState: LogitsProcessor (mod.rs:373-374) on the device.
// on == 0: ArgMax.  Otherwise Sampling::All { temperature }: the u32 words of the seeded ChaCha12 stream
// (rand 0.8 StdRng) consumed so far are counted here, so a captured decode graph draws the next word.
struct SampleState {
    uint32_t on;
    float inv_temp;          // (f32)(1 / temperature): `logits / temperature` is an affine multiply in candle
    uint32_t draw_lo, draw_hi;
    uint32_t key[8];
};

// acc += a.lo * b.lo + a.hi * b.hi on packed bf16 pairs (gfx950 VOP2; hipcc has no selectable builtin for it)
__device__ inline float dot2c_bf16(unsigned a, unsigned b, float acc) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}
// gfx940+: a VALU instruction that is not the same dot may read a dot's result only three wait states later (the next
// dot accumulates into it without any).  hipcc inserts such nops for its own instructions but cannot see into inline
// asm, so every run of dots ends with this (found as wrong sums of the LAST row of a block whose dots ran straight
// into the wave reduction).
__device__ inline void dot2c_settle(float &acc) { asm volatile("s_nop 2" : "+v"(acc)); }

// rotate-half RoPE of one pair: separately rounded products, as candle's mul / sub / add kernels produce them -- and no
// contraction into an FMA that a compiler picks one way in this build and another in the next (greedy token streams
// differed between two builds whose only change was elsewhere in the kernel)
__device__ inline void rope_rotate(float x0, float x1, float c, float s, float &t0, float &t1) {
    t0 = __fsub_rn(__fmul_rn(x0, c), __fmul_rn(x1, s));
    t1 = __fadd_rn(__fmul_rn(x0, s), __fmul_rn(x1, c));
}

}  // namespace fl
