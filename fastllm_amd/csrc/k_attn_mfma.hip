// k_attn_mfma.hip -- bf16 attention on the gfx950 matrix cores (v_mfma_f32_16x16x32_bf16), for the
// KV-cached single sequence of this path: decode (one token, G = H/Hkv query heads per kv head) and
// prefill (16 query tokens of one head per wave).  Flash-style: scores never leave registers.
//
// One wave processes 32 keys per step with 16 "query columns" (decode: the G heads of the kv group,
// padded to 16; prefill: 16 consecutive tokens of one head):
//   S^T = K . Q^T   A = K rows straight from the cache [key][d] (16 B per lane), B = Q fragments kept
//                   in registers; the 16 MFMA rows of the two row-tiles are mapped to keys as
//                   key = base + 8*(i>>2) + 4*tile + (i&3), so that afterwards lane (col q, group g4)
//                   holds the scores of keys base + 8*g4 + 0..7 -- exactly the A-operand layout of ...
//   O  += P . V     A = P (softmaxed scores, packed to bf16 in registers, no LDS, no transposition),
//                   B = V^T read 16 B per lane from the TRANSPOSED value cache [d][key].
// The value cache is therefore stored transposed in bf16 mode ([Hkv][d][seq_alloc]); the append of one
// token is d strided 2-byte stores (done by the QKV kernel's epilogue), a prefill appends rows of keys.
// Online softmax runs per column on the lanes that own it (max over the lane's 8 keys + two xor
// shuffles across the four lane groups); the rescale factor reaches the O accumulators (whose rows
// are columns of S^T) through four wave shuffles.
//
// Decode: grid (Hkv, nsplit), 4 waves per workgroup, 32-key tiles round-robin over the waves; the
// cross-wave and cross-split merge is shared with the VALU kernel (attn_common.h).
// Prefill: each wave owns one (16-token tile, head); causal + sliding-window mask per column.
#include "attn_common.h"

namespace fl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline bf16x8 ld_bf16x8(const bf16_t *p) { return *reinterpret_cast<const bf16x8 *>(p); }

template <int D>
struct MfmaAttnState {
    float m, l;                 // of column q = lane & 15 (m identical in the four lane groups, l partial)
    float4v O[D / 16];          // rows q = 4*(lane>>4) + r, column d = db*16 + (lane & 15)
    __device__ void init() {
        m = -INFINITY; l = 0.f;
#pragma unroll
        for (int i = 0; i < D / 16; i++) O[i] = float4v{0.f, 0.f, 0.f, 0.f};
    }
};

// One 32-key step.  Column q sees key k iff  k < pre_hi  ||  (lo_q <= k && k < hi_q).
template <int D>
__device__ inline void attn_tile(MfmaAttnState<D> &s, const bf16x8 (&qf)[D / 32], const bf16_t *__restrict__ kb,
                                 const bf16_t *__restrict__ vT, int ldv, int kbase, int pre_hi, int lo_q, int hi_q,
                                 float scale, int lane) {
    const int i = lane & 15, g4 = lane >> 4;
    const int key0 = kbase + 8 * (i >> 2) + (i & 3);
    float4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dk = 0; dk < D / 32; dk++) {
        const bf16x8 a0 = ld_bf16x8(kb + (size_t)key0 * D + dk * 32 + g4 * 8);
        const bf16x8 a1 = ld_bf16x8(kb + (size_t)(key0 + 4) * D + dk * 32 + g4 * 8);
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[dk], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[dk], s1, 0, 0, 0);
    }
    // value fragments for this step: issued now, consumed after the softmax
    bf16x8 vb[D / 16];
#pragma unroll
    for (int db = 0; db < D / 16; db++) vb[db] = ld_bf16x8(vT + (size_t)(db * 16 + i) * ldv + kbase + 8 * g4);

    float p[8];
    float mt = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int key = kbase + 8 * g4 + j;
        const float sc = (j < 4 ? s0[j & 3] : s1[j & 3]) * scale;
        const bool ok = key < pre_hi || (key >= lo_q && key < hi_q);
        p[j] = ok ? sc : -INFINITY;
        mt = fmaxf(mt, p[j]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(s.m, mt);
    const bool dead = mn == -INFINITY;                         // this column has seen no visible key yet
    const float alpha = dead ? 1.0f : __expf(s.m - mn);
    float ps = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) { p[j] = dead ? 0.f : __expf(p[j] - mn); ps += p[j]; }
    s.l = s.l * alpha + ps;
    s.m = mn;
    bf16x8 pa;
#pragma unroll
    for (int j = 0; j < 8; j++) pa[j] = (__bf16)p[j];
    float ar[4];
#pragma unroll
    for (int r = 0; r < 4; r++) ar[r] = __shfl(alpha, 4 * g4 + r, 64);     // factor of the O rows this lane holds
#pragma unroll
    for (int db = 0; db < D / 16; db++) {
#pragma unroll
        for (int r = 0; r < 4; r++) s.O[db][r] *= ar[r];
        s.O[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, vb[db], s.O[db], 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------- decode
template <int D, int GMAX, int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_mfma_kernel(const bf16_t *__restrict__ q, const bf16_t *__restrict__ kc,
                                                               const bf16_t *__restrict__ vT, const StepState *__restrict__ st,
                                                               float *__restrict__ part_m, float *__restrict__ part_l,
                                                               float *__restrict__ part_o, unsigned *__restrict__ counters,
                                                               bf16_t *__restrict__ out, int H, int Hkv, int seq_alloc,
                                                               float scale, int nsplit) {
    __shared__ float lds[NW * GMAX * (D + 2)];
    __shared__ int is_last;
    const int hk = blockIdx.x, split = blockIdx.y;
    const int G = H / Hkv, hq0 = hk * G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g4 = lane >> 4;
    const int S = (int)st->len + 1;
    int per = (S + nsplit - 1) / nsplit;
    per = (per + 32 * NW - 1) / (32 * NW) * (32 * NW);
    const int lo = split * per, hi = min(S, lo + per);

    bf16x8 qf[D / 32];
#pragma unroll
    for (int dk = 0; dk < D / 32; dk++) {
        if (i < G) qf[dk] = ld_bf16x8(q + (size_t)(hq0 + i) * D + dk * 32 + g4 * 8);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) qf[dk][j] = (__bf16)0.f;
        }
    }
    MfmaAttnState<D> s; s.init();
    const bf16_t *kb = kc + (size_t)hk * seq_alloc * D;
    const bf16_t *vb = vT + (size_t)hk * D * seq_alloc;
    for (int kbase = lo + 32 * wave; kbase < hi; kbase += 32 * NW)
        attn_tile<D>(s, qf, kb, vb, seq_alloc, kbase, 0, lo, hi, scale, lane);

    // wave slab -> LDS in the shared format [wave][head][o[D], m, l]
    float lt = s.l;
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    constexpr int STR = D + 2;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int qq = 4 * g4 + r;
        if (qq < G) {
            float *p = lds + ((size_t)wave * GMAX + qq) * STR;
#pragma unroll
            for (int db = 0; db < D / 16; db++) p[db * 16 + i] = s.O[db][r];
        }
    }
    if (g4 == 0 && i < G) {
        float *p = lds + ((size_t)wave * GMAX + i) * STR;
        p[D] = s.m; p[D + 1] = lt;
    }
    __syncthreads();
    decode_tail<bf16_t, D, GMAX, NW>(lds, &is_last, G, hq0, hk, split, nsplit, part_m, part_l, part_o, counters, out);
}

template <int D, int GMAX, int NW>
static int launch_decode_mfma_t(Launcher &L, const void *q, const void *kc, const void *vT, const StepState *st, void *out,
                                const AttnScratch &sc, int64_t H, int64_t Hkv, int64_t seq_alloc, float scale) {
    dim3 grid((unsigned)Hkv, (unsigned)sc.nsplit, 1);
    double kvbytes = 2.0 * (double)sc.kv_len_hint * Hkv * D * 2;
    return L.launch(KC_ATTN_DECODE, kvbytes, 4.0 * (double)sc.kv_len_hint * H * D, attn_decode_mfma_kernel<D, GMAX, NW>, grid,
                    dim3(NW * 64), 0, (const bf16_t *)q, (const bf16_t *)kc, (const bf16_t *)vT, st, sc.part_m, sc.part_l,
                    sc.part_o, sc.counters, (bf16_t *)out, (int)H, (int)Hkv, (int)seq_alloc, scale, sc.nsplit);
}

bool attn_mfma_supported(int dtype, int64_t H, int64_t Hkv, int64_t d) {
    return dtype == FL_DTYPE_BF16 && (d == 64 || d == 128) && Hkv > 0 && H % Hkv == 0 && H / Hkv <= 16;
}

int launch_attn_decode_mfma(Launcher &L, const void *q, const void *k_cache, const void *v_cache_T, const StepState *st,
                            void *out, const AttnScratch &sc, int64_t H, int64_t Hkv, int64_t d, int64_t seq_alloc,
                            float scale) {
    const int G = (int)(H / Hkv);
    if (sc.nsplit > 64) FL_FAIL(FL_ERR_BAD_ARGUMENT, "attention: at most 64 splits");
    // nsplit == 1: one wide workgroup per kv head (no cross-workgroup combine); else 4-wave workgroups
    const bool wide = sc.nsplit == 1;
#define FL_GO(DD)                                                                                                   \
    if (G <= 4) return wide ? launch_decode_mfma_t<DD, 4, 16>(L, q, k_cache, v_cache_T, st, out, sc, H, Hkv, seq_alloc, scale)  \
                            : launch_decode_mfma_t<DD, 4, 4>(L, q, k_cache, v_cache_T, st, out, sc, H, Hkv, seq_alloc, scale);  \
    if (G <= 8) return wide ? launch_decode_mfma_t<DD, 8, 16>(L, q, k_cache, v_cache_T, st, out, sc, H, Hkv, seq_alloc, scale)  \
                            : launch_decode_mfma_t<DD, 8, 4>(L, q, k_cache, v_cache_T, st, out, sc, H, Hkv, seq_alloc, scale);  \
    return wide ? launch_decode_mfma_t<DD, 16, 8>(L, q, k_cache, v_cache_T, st, out, sc, H, Hkv, seq_alloc, scale)              \
                : launch_decode_mfma_t<DD, 16, 4>(L, q, k_cache, v_cache_T, st, out, sc, H, Hkv, seq_alloc, scale);
    if (d == 128) { FL_GO(128) }
    if (d == 64) { FL_GO(64) }
#undef FL_GO
    FL_FAIL(FL_ERR_UNSUPPORTED, "mfma attention: head_dim %lld", (long long)d);
}

// ------------------------------------------------------------------------------- prefill
// grid (ceil(T/16/4), H), 4 waves; wave w owns query tokens [t0, t0+16) of head blockIdx.y.
// Mask (App. A.5): cached prefix [0,len) visible; in-call key j visible to token t iff j <= t and
// (window < 0 or j + window >= t).
template <int D>
__global__ __launch_bounds__(256) void attn_prefill_mfma_kernel(const bf16_t *__restrict__ q, const bf16_t *__restrict__ kc,
                                                                const bf16_t *__restrict__ vT, const StepState *__restrict__ st,
                                                                bf16_t *__restrict__ out, int T, int H, int Hkv, int seq_alloc,
                                                                float scale, int window) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 15, g4 = lane >> 4;
    const int t0 = (blockIdx.x * 4 + wave) * 16;
    if (t0 >= T) return;                                         // wave-uniform
    const int hq = blockIdx.y, hk = hq / (H / Hkv);
    const int len = (int)st->len;
    const int t = t0 + i;                                        // this lane's column
    const bool col_ok = t < T;

    bf16x8 qf[D / 32];
#pragma unroll
    for (int dk = 0; dk < D / 32; dk++) {
        if (col_ok) qf[dk] = ld_bf16x8(q + ((size_t)t * H + hq) * D + dk * 32 + g4 * 8);
        else {
#pragma unroll
            for (int j = 0; j < 8; j++) qf[dk][j] = (__bf16)0.f;
        }
    }
    int jlo = 0;
    if (window >= 0 && t - window > 0) jlo = t - window;
    const int lo_q = col_ok ? len + jlo : 0, hi_q = col_ok ? len + t + 1 : 0;
    const int pre_hi = col_ok ? len : 0;

    MfmaAttnState<D> s; s.init();
    const bf16_t *kb = kc + (size_t)hk * seq_alloc * D;
    const bf16_t *vb = vT + (size_t)hk * D * seq_alloc;
    // the tile range of the whole wave: from the first key any column can see to the last token's key
    int kstart = 0;
    if (len == 0 && window >= 0 && t0 - window > 0) kstart = ((t0 - window) / 32) * 32;
    const int kend = len + min(T, t0 + 16);
    for (int kbase = kstart; kbase < kend; kbase += 32)
        attn_tile<D>(s, qf, kb, vb, seq_alloc, kbase, pre_hi, lo_q, hi_q, scale, lane);

    float lt = s.l;
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int qq = 4 * g4 + r;
        const float lr = __shfl(lt, qq, 64);
        const int tt = t0 + qq;
        if (tt < T) {
            const float inv = 1.0f / lr;
            bf16_t *o = out + ((size_t)tt * H + hq) * D;
#pragma unroll
            for (int db = 0; db < D / 16; db++) o[db * 16 + i] = float_to_bf16_bits(s.O[db][r] * inv);
        }
    }
}

int launch_attn_prefill_mfma(Launcher &L, const void *q, const void *k_cache, const void *v_cache_T, const StepState *st,
                             void *out, int64_t T, int64_t H, int64_t Hkv, int64_t d, int64_t seq_alloc, float scale,
                             int64_t window) {
    dim3 grid((unsigned)((T + 63) / 64), (unsigned)H);
    double flops = 2.0 * (double)T * T * H * d;
    if (d == 128)
        return L.launch(KC_ATTN_PREFILL, 0, flops, attn_prefill_mfma_kernel<128>, grid, dim3(256), 0, (const bf16_t *)q,
                        (const bf16_t *)k_cache, (const bf16_t *)v_cache_T, st, (bf16_t *)out, (int)T, (int)H, (int)Hkv,
                        (int)seq_alloc, scale, (int)window);
    if (d == 64)
        return L.launch(KC_ATTN_PREFILL, 0, flops, attn_prefill_mfma_kernel<64>, grid, dim3(256), 0, (const bf16_t *)q,
                        (const bf16_t *)k_cache, (const bf16_t *)v_cache_T, st, (bf16_t *)out, (int)T, (int)H, (int)Hkv,
                        (int)seq_alloc, scale, (int)window);
    FL_FAIL(FL_ERR_UNSUPPORTED, "mfma attention: head_dim %lld", (long long)d);
}

}  // namespace fl
