// k_ops.hip -- the small memory-bound ops of the decoder: embedding gather (K1), fused
// residual-add + RMSNorm (K2/K9), RoPE + KV-cache append (K4/K5), device argmax (K13),
// local shard reduction, and the one-off weight conversion at model build.
#include <stdlib.h>

#include "kernels.h"

namespace fl {

static const char *kNames[KC_COUNT] = {
    "embed", "rmsnorm_add", "gemv", "gemm_mfma", "gemm_generic", "rope_kv_append", "attn_decode",
    "attn_combine", "attn_prefill", "select_advance", "reduce_shards", "convert", "attn_oproj", "comm_oneshot"};
const char *kernel_class_name(int kc) { return (kc >= 0 && kc < KC_COUNT) ? kNames[kc] : "?"; }

// ------------------------------------------------------------------------------- embedding
// Embedding::forward (index_select): x[t,:] = E[ids[t],:]; residual stream is fp32.
template <typename WT>
__global__ __launch_bounds__(256) void embed_kernel(const WT *__restrict__ E, const uint32_t *__restrict__ ids,
                                                    const StepState *__restrict__ st, float *__restrict__ x, int h) {
    const int t = blockIdx.x;
    const uint32_t id = ids ? ids[t] : st->token;
    const WT *row = E + (size_t)id * h;
    float *dst = x + (size_t)t * h;
    for (int c = threadIdx.x; c * 8 < h; c += 256) {
        float v[8];
        load8(row + c * 8, v);
        store8(dst + c * 8, v);
    }
}

int launch_embed(Launcher &L, int dtype, const void *E, const uint32_t *ids, const StepState *st,
                 float *x_res, int64_t T, int64_t h) {
    double bytes = (double)T * h * ((dtype == FL_DTYPE_BF16 ? 2 : 4) + 4);
    if (dtype == FL_DTYPE_BF16)
        return L.launch(KC_EMBED, bytes, 0, embed_kernel<bf16_t>, dim3((unsigned)T), dim3(256), 0,
                        (const bf16_t *)E, ids, st, x_res, (int)h);
    return L.launch(KC_EMBED, bytes, 0, embed_kernel<float>, dim3((unsigned)T), dim3(256), 0,
                    (const float *)E, ids, st, x_res, (int)h);
}

// ------------------------------------------------------------------------------- rmsnorm (+add)
// candle_nn::ops::rms_norm (App. A.2): m = sqrt(sum(x^2)/h + eps), y = x / m * w, sum in fp32.
// Fused with the preceding residual add (x_res += delta) so the residual stream is read once.
// Emits xs = x * w and inv_rms = 1/m separately: y = inv_rms * xs is finished by the consumer's
// epilogue (single pass over x here, and the same rounding point as the fused decode prologue).
template <typename OT>
__global__ __launch_bounds__(256) void rmsnorm_add_kernel(float *__restrict__ x_res, const float *__restrict__ delta,
                                                          const float *__restrict__ w, float eps,
                                                          OT *__restrict__ xs, float *__restrict__ inv_rms, int h,
                                                          int n_slab, long long slab_stride) {
    __shared__ float red[4];
    const int t = blockIdx.x, tid = threadIdx.x;
    float *xr = x_res + (size_t)t * h;
    const float *dr = delta ? delta + (size_t)t * h : nullptr;
    float ss = 0.f;
    for (int c = tid; c * 8 < h; c += 256) {
        float v[8], wv[8], o[8];
        load8(xr + c * 8, v);
        load8(w + c * 8, wv);
        if (dr) {
            // split-K slabs, summed in slab order; four are requested at a time (one slab per round trip made this launch a
            // latency chain at eight slabs: Mistral-7B T = 512 13.3 us)
            for (int sl = 0; sl < n_slab; sl += 4) {
                float d[4][8];
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (sl + u < n_slab) load8(dr + (size_t)(sl + u) * slab_stride + c * 8, d[u]);
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (sl + u < n_slab) {
#pragma unroll
                        for (int j = 0; j < 8; j++) v[j] += d[u][j];
                    }
            }
            store8(xr + c * 8, v);
        }
#pragma unroll
        for (int j = 0; j < 8; j++) { ss = fmaf(v[j], v[j], ss); o[j] = v[j] * wv[j]; }
        store8(xs + (size_t)t * h + c * 8, o);
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) inv_rms[t] = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)h + eps);
}

int launch_rmsnorm_add(Launcher &L, int dtype, float *x_res, const float *delta, const float *w, float eps,
                       void *xs, float *inv_rms, int64_t T, int64_t h, int n_slab, int64_t slab_stride) {
    double bytes = (double)T * h * (4.0 * (delta ? 2 + n_slab : 1) + (dtype == FL_DTYPE_BF16 ? 2 : 4));
    if (dtype == FL_DTYPE_BF16)
        return L.launch(KC_RMSNORM, bytes, 0, rmsnorm_add_kernel<bf16_t>, dim3((unsigned)T), dim3(256), 0,
                        x_res, delta, w, eps, (bf16_t *)xs, inv_rms, (int)h, n_slab, (long long)slab_stride);
    return L.launch(KC_RMSNORM, bytes, 0, rmsnorm_add_kernel<float>, dim3((unsigned)T), dim3(256), 0,
                    x_res, delta, w, eps, (float *)xs, inv_rms, (int)h, n_slab, (long long)slab_stride);
}

// 1/rms from the partial sums of squares the EPI_RESID GEMM epilogue left (kernels.h): one wave per row, fixed tree order
__global__ __launch_bounds__(256) void rms_finalize_kernel(const float *__restrict__ part, int np, float eps, float *__restrict__ inv_rms, int T, int h) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    float ss = 0.f;
    for (int i = lane; i < np; i += 64) ss += part[(size_t)t * np + i];
    ss = wave_sum(ss);
    if (lane == 0) inv_rms[t] = 1.0f / sqrtf(ss / (float)h + eps);
}
int rs_parts_to_vector(Launcher &L, const float *row_scale, int64_t T) {
    if (!L.rsp.part) return FL_OK;
    if (!row_scale || !(L.rsp.inv_h > 0.f)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "row scales left as partial sums, and no vector to finish them into");
    const RsParts p = L.rsp;
    L.rsp = RsParts{};
    return launch_rms_finalize(L, p.part, p.np, p.eps, const_cast<float *>(row_scale), T, (int64_t)(1.0f / p.inv_h + 0.5f));
}

int launch_rms_finalize(Launcher &L, const float *part, int np, float eps, float *inv_rms, int64_t T, int64_t h) {
    Launcher LL = L; LL.tag = "finalize";
    return LL.launch(KC_RMSNORM, (double)T * np * 4.0, 0, rms_finalize_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, part, np, eps, inv_rms,
                     (int)T, (int)h);
}

// ------------------------------------------------------------------------------- RoPE + KV append
// candle_nn::rotary_emb::rope (rotate-half, App. A.4) on q and k, then the new K/V rows are
// written in place at cache index len+t (the reference's Tensor::cat copies the whole cache every
// step, K5).  One thread per (t, head, j < d/2) pair.  cos/sin come from host-built fp32 tables.
template <typename CT>
__global__ __launch_bounds__(256) void rope_kv_kernel(const float *__restrict__ qkv, const StepState *__restrict__ st,
                                                      const float *__restrict__ cos_tab, const float *__restrict__ sin_tab,
                                                      int max_pos, CT *__restrict__ q_out, CT *__restrict__ kc,
                                                      CT *__restrict__ vc, int T, int H, int Hkv, int d, int max_seq,
                                                      int v_transposed, int nslab, long long slab_stride,
                                                      const float *__restrict__ bias) {
    const int half = d >> 1;
    const int nheads = H + 2 * Hkv;
    const int per_t = nheads * half;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)T * per_t) return;
    const int t = (int)(idx / per_t), rem = (int)(idx % per_t);
    const int hd = rem / half, j = rem % half;
    const float *src = qkv + (size_t)t * nheads * d + (size_t)hd * d;
    float a = src[j], b = src[j + half];
    for (int sl = 1; sl < nslab; sl++) {             // split-K slabs of the projection, summed in slab order
        a += src[(size_t)sl * slab_stride + j];
        b += src[(size_t)sl * slab_stride + j + half];
    }
    if (bias) { a += bias[(size_t)hd * d + j]; b += bias[(size_t)hd * d + j + half]; }     // (the projection ran in K slices: its bias comes here)
    const uint32_t pos = st->pos + (uint32_t)t, slot = st->len + (uint32_t)t;
    if (hd < H + Hkv) {
        const uint32_t p = pos < (uint32_t)max_pos ? pos : (uint32_t)max_pos - 1;   // host validates range
        const float c = cos_tab[(size_t)p * half + j], s = sin_tab[(size_t)p * half + j];
        float ra, rb;
        rope_rotate(a, b, c, s, ra, rb);
        if (hd < H) {
            CT *o = q_out + ((size_t)t * H + hd) * d;
            elem<CT>::st(o + j, ra); elem<CT>::st(o + j + half, rb);
        } else {
            CT *o = kc + ((size_t)(hd - H) * max_seq + slot) * d;
            elem<CT>::st(o + j, ra); elem<CT>::st(o + j + half, rb);
        }
    } else if (v_transposed) {                     // [Hkv][d][max_seq]: one key column per token
        CT *o = vc + (size_t)(hd - H - Hkv) * d * max_seq + slot;
        elem<CT>::st(o + (size_t)j * max_seq, a); elem<CT>::st(o + (size_t)(j + half) * max_seq, b);
    } else {
        CT *o = vc + ((size_t)(hd - H - Hkv) * max_seq + slot) * d;
        elem<CT>::st(o + j, a); elem<CT>::st(o + j + half, b);
    }
}

// The prompt-sized form (bf16, transposed value cache, T > 1): 8 pairs per thread with 16-byte loads / stores for q and k,
// and V through a 32-token x 64-row LDS transpose so that the value cache [Hkv][d][seq] receives 64-byte runs of
// consecutive tokens instead of one strided 2-byte store per element (Mistral-7B T = 512: 11 -> ~6 us per layer).
// Blocks [0, nA): 256 (token, q-or-k head, 8-pair chunk) items each; blocks [nA, ..): one V tile each.
__global__ __launch_bounds__(256) void rope_kv_vec_kernel(const float *__restrict__ qkv, const StepState *__restrict__ st,
                                                          const float *__restrict__ cos_tab, const float *__restrict__ sin_tab,
                                                          int max_pos, bf16_t *__restrict__ q_out, bf16_t *__restrict__ kc,
                                                          bf16_t *__restrict__ vc, int T, int H, int Hkv, int d, int max_seq,
                                                          int nslab, long long slab_stride, int nA, const float *__restrict__ bias) {
    __shared__ bf16_t tile[64][40];                       // [d row][token], 80-byte rows: 16-byte aligned chunks of 8 tokens
    const int half = d >> 1, nheads = H + 2 * Hkv, tid = threadIdx.x;
    const uint32_t len = st->len, pos0 = st->pos;
    if ((int)blockIdx.x < nA) {
        const int cph = half >> 3;                        // 8-pair chunks per head
        const long long idx = (long long)blockIdx.x * 256 + tid;
        if (idx >= (long long)T * (H + Hkv) * cph) return;
        const int t = (int)(idx / ((H + Hkv) * cph)), rem = (int)(idx % ((H + Hkv) * cph));
        const int hd = rem / cph, j0 = (rem % cph) * 8;
        const float *src = qkv + (size_t)t * nheads * d + (size_t)hd * d + j0;
        float a[8], b[8];
        load8(src, a); load8(src + half, b);
        for (int sl = 1; sl < nslab; sl++) {              // split-K slabs of the projection, summed in slab order
            float a2[8], b2[8];
            load8(src + (size_t)sl * slab_stride, a2); load8(src + (size_t)sl * slab_stride + half, b2);
#pragma unroll
            for (int j = 0; j < 8; j++) { a[j] += a2[j]; b[j] += b2[j]; }
        }
        if (bias) {
            float ba[8], bb[8];
            load8(bias + (size_t)hd * d + j0, ba); load8(bias + (size_t)hd * d + j0 + half, bb);
#pragma unroll
            for (int j = 0; j < 8; j++) { a[j] += ba[j]; b[j] += bb[j]; }
        }
        const uint32_t pos = pos0 + (uint32_t)t, p = pos < (uint32_t)max_pos ? pos : (uint32_t)max_pos - 1;
        float c[8], sn[8], ra[8], rb[8];
        load8(cos_tab + (size_t)p * half + j0, c); load8(sin_tab + (size_t)p * half + j0, sn);
#pragma unroll
        for (int j = 0; j < 8; j++) rope_rotate(a[j], b[j], c[j], sn[j], ra[j], rb[j]);     // (separately rounded products, as every other RoPE site)
        bf16_t *o = hd < H ? q_out + ((size_t)t * H + hd) * d + j0 : kc + ((size_t)(hd - H) * max_seq + len + t) * d + j0;
        store8(o, ra); store8(o + half, rb);
        return;
    }
    // ---- V tile: 32 tokens x 64 d rows of one kv head ----
    const int vb = (int)blockIdx.x - nA, dpt = d / 64;    // d tiles per head
    const int tt = vb / (Hkv * dpt), hk = (vb / dpt) % Hkv, d0 = (vb % dpt) * 64, t0 = tt * 32;
    {
        const int tl = tid >> 3, t = t0 + tl, j0 = (tid & 7) * 8;
        if (t < T) {
            const float *src = qkv + (size_t)t * nheads * d + (size_t)(H + Hkv + hk) * d + d0 + j0;
            float v[8];
            load8(src, v);
            for (int sl = 1; sl < nslab; sl++) {
                float v2[8];
                load8(src + (size_t)sl * slab_stride, v2);
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] += v2[j];
            }
            if (bias) {
                float bv[8];
                load8(bias + (size_t)(H + Hkv + hk) * d + d0 + j0, bv);
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] += bv[j];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) tile[j0 + j][tl] = float_to_bf16_bits(v[j]);
        }
    }
    __syncthreads();
    const int r = tid >> 2, c8 = (tid & 3) * 8;           // d row, first of 8 tokens
    bf16_t *dst = vc + ((size_t)hk * d + d0 + r) * max_seq + len + t0 + c8;
    if (t0 + c8 + 8 <= T && ((len + t0 + c8) & 7) == 0) {
        *reinterpret_cast<uint4v *>(dst) = *reinterpret_cast<const uint4v *>(&tile[r][c8]);
    } else {
        for (int j = 0; j < 8 && t0 + c8 + j < T; j++) dst[j] = tile[r][c8 + j];
    }
}

int launch_rope_kv(Launcher &L, int dtype, const float *qkv, const StepState *st, const float *cos_tab,
                   const float *sin_tab, int64_t max_pos, void *q_out, void *k_cache, void *v_cache,
                   int64_t T, int64_t H, int64_t Hkv, int64_t d, int64_t max_seq, bool v_transposed, int nslab, const float *bias) {
    const long long slab_stride = (long long)T * (H + 2 * Hkv) * d;
    const int64_t total = T * (H + 2 * Hkv) * (d / 2);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    const int es = dtype == FL_DTYPE_BF16 ? 2 : 4;
    double bytes = (double)T * (H + 2 * Hkv) * d * (4 + es);
    const int use_vec = tune(TK_ROPE_VEC);
    if (use_vec && dtype == FL_DTYPE_BF16 && v_transposed && T >= 16 && d % 64 == 0 && max_seq % 8 == 0) {
        const int64_t itemsA = T * (H + Hkv) * (d / 16);
        const int nA = (int)((itemsA + 255) / 256), nB = (int)(((T + 31) / 32) * Hkv * (d / 64));
        return L.launch(KC_ROPE_KV, bytes, 0, rope_kv_vec_kernel, dim3((unsigned)(nA + nB)), dim3(256), 0, qkv, st, cos_tab, sin_tab,
                        (int)max_pos, (bf16_t *)q_out, (bf16_t *)k_cache, (bf16_t *)v_cache, (int)T, (int)H, (int)Hkv, (int)d,
                        (int)max_seq, nslab, slab_stride, nA, bias);
    }
    if (dtype == FL_DTYPE_BF16)
        return L.launch(KC_ROPE_KV, bytes, 0, rope_kv_kernel<bf16_t>, dim3(blocks), dim3(256), 0, qkv, st, cos_tab,
                        sin_tab, (int)max_pos, (bf16_t *)q_out, (bf16_t *)k_cache, (bf16_t *)v_cache, (int)T, (int)H,
                        (int)Hkv, (int)d, (int)max_seq, (int)v_transposed, nslab, slab_stride, bias);
    return L.launch(KC_ROPE_KV, bytes, 0, rope_kv_kernel<float>, dim3(blocks), dim3(256), 0, qkv, st, cos_tab,
                    sin_tab, (int)max_pos, (float *)q_out, (float *)k_cache, (float *)v_cache, (int)T, (int)H,
                    (int)Hkv, (int)d, (int)max_seq, (int)v_transposed, nslab, slab_stride, bias);
}

// ------------------------------------------------------------------------------- batched variants (row N4)
// One row per sequence of a batch: the token, RoPE position, KV slot and caches come from its SeqRef.
template <typename WT>
__global__ __launch_bounds__(256) void embed_batch_kernel(const WT *__restrict__ E, const SeqRef *__restrict__ seqs,
                                                          float *__restrict__ x, int h) {
    const int b = blockIdx.x;
    const WT *row = E + (size_t)seqs[b].st->token * h;
    float *dst = x + (size_t)b * h;
    for (int c = threadIdx.x; c * 8 < h; c += 256) {
        float v[8];
        load8(row + c * 8, v);
        store8(dst + c * 8, v);
    }
}

int launch_embed_batch(Launcher &L, const void *E, const SeqRef *seqs_dev, float *x_res, int B, int64_t h, int dtype) {
    if (dtype == FL_DTYPE_F32)
        return L.launch(KC_EMBED, (double)B * h * 8, 0, embed_batch_kernel<float>, dim3((unsigned)B), dim3(256), 0, (const float *)E, seqs_dev, x_res, (int)h);
    return L.launch(KC_EMBED, (double)B * h * 6, 0, embed_batch_kernel<bf16_t>, dim3((unsigned)B), dim3(256), 0, (const bf16_t *)E, seqs_dev,
                    x_res, (int)h);
}

// qkv fp32 [B][(H+2Hkv)*d] -> RoPE -> q bf16 [B][H*d], rotated k and v appended to each sequence's own cache
// (V transposed, the MFMA attention layout); kv_layer_off = layer * Hkv * d, times the sequence's seq_alloc
template <typename CT, bool VT>
__global__ __launch_bounds__(256) void rope_kv_batch_kernel(const float *__restrict__ qkv, const SeqRef *__restrict__ seqs,
                                                            const float *__restrict__ cos_tab, const float *__restrict__ sin_tab,
                                                            int max_pos, CT *__restrict__ q_out, size_t kv_layer_off, int B,
                                                            int H, int Hkv, int d, int n_slab, const float *__restrict__ bias) {
    const int half = d >> 1;
    const int nheads = H + 2 * Hkv;
    const int per_t = nheads * half;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)B * per_t) return;
    const int b = (int)(idx / per_t), rem = (int)(idx % per_t);
    const int hd = rem / half, j = rem % half;
    const SeqRef &sq = seqs[b];
    const float *src = qkv + (size_t)b * nheads * d + (size_t)hd * d;
    float a = src[j], bb = src[j + half];
    for (int s = 1; s < n_slab; s++) {                    // K slices of the projection, summed in slab order
        const float *ss = src + (size_t)s * B * nheads * d;
        a += ss[j]; bb += ss[j + half];
    }
    if (bias) { a += bias[hd * d + j]; bb += bias[hd * d + j + half]; }
    const uint32_t pos = sq.st->pos, slot = sq.st->len;
    const size_t sa = (size_t)sq.seq_alloc;
    if (hd < H + Hkv) {
        const uint32_t p = pos < (uint32_t)max_pos ? pos : (uint32_t)max_pos - 1;
        const float c = cos_tab[(size_t)p * half + j], s = sin_tab[(size_t)p * half + j];
        float ra, rb;
        rope_rotate(a, bb, c, s, ra, rb);
        CT *o = hd < H ? q_out + ((size_t)b * H + hd) * d
                       : reinterpret_cast<CT *>(sq.k) + kv_layer_off * sa + ((size_t)(hd - H) * sa + slot) * d;
        elem<CT>::st(o + j, ra); elem<CT>::st(o + j + half, rb);
    } else if (VT) {                                      // V transposed [Hkv][d][seq_alloc]: the MFMA attention layout
        CT *o = reinterpret_cast<CT *>(sq.v) + kv_layer_off * sa + (size_t)(hd - H - Hkv) * d * sa + slot;
        elem<CT>::st(o + (size_t)j * sa, a); elem<CT>::st(o + (size_t)(j + half) * sa, bb);
    } else {                                              // V [Hkv][seq_alloc][d]
        CT *o = reinterpret_cast<CT *>(sq.v) + kv_layer_off * sa + ((size_t)(hd - H - Hkv) * sa + slot) * d;
        elem<CT>::st(o + j, a); elem<CT>::st(o + j + half, bb);
    }
}

int launch_rope_kv_batch(Launcher &L, const float *qkv, const SeqRef *seqs_dev, const float *cos_tab, const float *sin_tab,
                         int64_t max_pos, void *q_out, size_t kv_layer_off, int B, int64_t H, int64_t Hkv, int64_t d, int n_slab,
                         const float *bias, int dtype, bool v_transposed) {
    const int64_t total = (int64_t)B * (H + 2 * Hkv) * (d / 2);
    const dim3 grid((unsigned)((total + 255) / 256));
    const double bytes = (double)B * (H + 2 * Hkv) * d * (dtype == FL_DTYPE_F32 ? 8 : 6);
    if (dtype == FL_DTYPE_F32)
        return L.launch(KC_ROPE_KV, bytes, 0, rope_kv_batch_kernel<float, false>, grid, dim3(256), 0, qkv, seqs_dev, cos_tab, sin_tab, (int)max_pos,
                        (float *)q_out, kv_layer_off, B, (int)H, (int)Hkv, (int)d, n_slab, bias);
    if (!v_transposed)
        return L.launch(KC_ROPE_KV, bytes, 0, rope_kv_batch_kernel<bf16_t, false>, grid, dim3(256), 0, qkv, seqs_dev, cos_tab, sin_tab, (int)max_pos,
                        (bf16_t *)q_out, kv_layer_off, B, (int)H, (int)Hkv, (int)d, n_slab, bias);
    return L.launch(KC_ROPE_KV, bytes, 0, rope_kv_batch_kernel<bf16_t, true>, grid, dim3(256), 0, qkv, seqs_dev, cos_tab, sin_tab, (int)max_pos,
                    (bf16_t *)q_out, kv_layer_off, B, (int)H, (int)Hkv, (int)d, n_slab, bias);
}

// ------------------------------------------------------------------------------- token selection + advance
// LogitsProcessor::sample (mod.rs:308-310,425-428) on the device, then the loop-carried state of
// mod.rs:411-453 (token, pos, len, step, eos) is advanced so that a captured decode graph can be
// replayed back to back with no host round trip.
//
// a workgroup's (value, index) pairs -> the ArgMax in thread 0 (ties -> the larger index; index < 0 = nothing)
__device__ inline int argmax_reduce(float best, int idx, float *bv, int *bi) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64); int oi = __shfl_xor(idx, o, 64);
        if (oi >= 0 && (idx < 0 || ov > best || (ov == best && oi > idx))) { best = ov; idx = oi; }
    }
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; w++) {
            float ov = bv[w]; int oi = bi[w];
            if (oi >= 0 && (idx < 0 || ov > best || (ov == best && oi > idx))) { best = ov; idx = oi; }
        }
    }
    return idx < 0 ? 0 : idx;                           // valid in thread 0
}

// ArgMax (temperature None or < 1e-7, App. A.7): iter().enumerate().max_by(total_cmp) -> on exact ties
// the LAST maximal index wins.
__device__ inline int argmax_last(const float *__restrict__ logits, int V, float *bv, int *bi) {
    const int tid = threadIdx.x;
    float best = -INFINITY; int idx = -1;
    auto take = [&](float v, int ii) { if (idx < 0 || v > best || (v == best && ii > idx)) { best = v; idx = ii; } };
    // 16 bytes per lane, eight loads per round trip, all unconditional (past the end a thread re-reads the last chunk and
    // skips it): V = 32000 is ONE round trip.  (Scalar loads with a one-load-per-iteration remainder loop: eleven.)
    const int nvec = (V % 4 == 0 && (reinterpret_cast<uintptr_t>(logits) & 15) == 0) ? V / 4 : 0;
    for (int c0 = tid; c0 < nvec; c0 += 8 * 1024) {
        float4v v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const float4v *>(logits + 4 * (size_t)min(c0 + u * 1024, nvec - 1));
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int c = c0 + u * 1024;
            if (c < nvec) {
#pragma unroll
                for (int j = 0; j < 4; j++) take(v[u][j], 4 * c + j);
            }
        }
    }
    for (int i = 4 * nvec + tid; i < V; i += 1024) take(logits[i], i);      // (a vocabulary that is not a multiple of 4)
    return argmax_reduce(best, idx, bv, bi);
}

// ArgMax over the candidates the lm_head launch left (GemvArgs::amax: one per workgroup, ties -> the larger index):
// the same result as argmax_last over the logits they were taken from, without reading the vocabulary again
__device__ inline int argmax_candidates(const ArgmaxCand *__restrict__ cand, float *bv, int *bi) {
    const int n = cand[0].i, tid = threadIdx.x;
    float best = -INFINITY; int idx = -1;
    if (tid < n) { const ArgmaxCand c = cand[1 + tid]; best = c.v; idx = c.i; }
    return argmax_reduce(best, idx, bv, bi);
}

// rand_chacha ChaCha12 block `counter` (64-bit block counter, stream id 0) of the stream keyed by `key`
__device__ inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
__device__ inline uint32_t chacha12_word(const uint32_t *key, uint64_t word_index) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
#pragma unroll
    for (int i = 0; i < 8; i++) s[4 + i] = key[i];
    const uint64_t counter = word_index >> 4;
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = 0; s[15] = 0;
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] = s[i];
#define FL_QR(a, b, c, d) \
    a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); \
    a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7);
#pragma unroll
    for (int r = 0; r < 12; r += 2) {
        FL_QR(x[0], x[4], x[8], x[12])  FL_QR(x[1], x[5], x[9], x[13])
        FL_QR(x[2], x[6], x[10], x[14]) FL_QR(x[3], x[7], x[11], x[15])
        FL_QR(x[0], x[5], x[10], x[15]) FL_QR(x[1], x[6], x[11], x[12])
        FL_QR(x[2], x[7], x[8], x[13])  FL_QR(x[3], x[4], x[9], x[14])
    }
#undef FL_QR
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) if ((int)(word_index & 15) == i) out = x[i] + s[i];
    return out;
}

// Left-to-right fp32 sum of p[0..V) by ONE lane -- the order of the reference's scalar loops -- with the
// other 1023 threads staging tiles of p through LDS ahead of it.  cumulative: p[i] is replaced by the
// running sum p[0]+..+p[i] (rand's WeightedIndex::new keeps exactly those).
constexpr int kSelTile = 4096;
__device__ inline float sequential_sum(float *__restrict__ p, int V, float *tiles /* 2 x kSelTile */, float *bcast, bool cumulative) {
    const int tid = threadIdx.x;
    const int ntiles = (V + kSelTile - 1) / kSelTile;
    float acc = 0.f;
    auto load_tile = [&](int t) {
        float *buf = tiles + (t & 1) * kSelTile;
        for (int j = tid; j < kSelTile; j += 1024) { const int i = t * kSelTile + j; buf[j] = i < V ? p[i] : 0.f; }
    };
    auto store_tile = [&](int t) {
        const float *buf = tiles + (t & 1) * kSelTile;
        for (int j = tid; j < kSelTile; j += 1024) { const int i = t * kSelTile + j; if (i < V) p[i] = buf[j]; }
    };
    load_tile(0);
    __syncthreads();
    for (int t = 0; t < ntiles; t++) {
        if (tid == 0) {
            float *buf = tiles + (t & 1) * kSelTile;
            const int n = min(kSelTile, V - t * kSelTile);
            // one dependent add chain; the LDS reads of the next 16 values are issued before the adds of
            // the current 16 so that their latency hides behind the chain
            int j = 0;
            float4v cur[4], nxt[4];
            if (n >= 16) for (int u = 0; u < 4; u++) cur[u] = *reinterpret_cast<const float4v *>(buf + 4 * u);
            for (; j + 16 <= n; j += 16) {
                const bool more = j + 32 <= n;
                if (more) for (int u = 0; u < 4; u++) nxt[u] = *reinterpret_cast<const float4v *>(buf + j + 16 + 4 * u);
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    float4v c;
                    c[0] = acc = __fadd_rn(acc, cur[u][0]); c[1] = acc = __fadd_rn(acc, cur[u][1]);
                    c[2] = acc = __fadd_rn(acc, cur[u][2]); c[3] = acc = __fadd_rn(acc, cur[u][3]);
                    if (cumulative) *reinterpret_cast<float4v *>(buf + j + 4 * u) = c;
                }
                if (more) for (int u = 0; u < 4; u++) cur[u] = nxt[u];
            }
            for (; j < n; j++) { acc = __fadd_rn(acc, buf[j]); if (cumulative) buf[j] = acc; }
        } else if (tid >= 64) {                      // the summing wave does nothing else
            // buffer (t+1)&1 held tile t-1: write its running sums back, then refill it with tile t+1;
            // a thread touches the same slots in both steps, so no barrier is needed between them
            if (cumulative && t >= 1) {
                const float *buf = tiles + ((t - 1) & 1) * kSelTile;
                for (int j = tid - 64; j < kSelTile; j += 960) { const int i = (t - 1) * kSelTile + j; if (i < V) p[i] = buf[j]; }
            }
            if (t + 1 < ntiles) {
                float *buf = tiles + ((t + 1) & 1) * kSelTile;
                for (int j = tid - 64; j < kSelTile; j += 960) { const int i = (t + 1) * kSelTile + j; buf[j] = i < V ? p[i] : 0.f; }
            }
        }
        __syncthreads();
    }
    if (cumulative) store_tile(ntiles - 1);
    if (tid == 0) bcast[0] = acc;
    __syncthreads();
    return bcast[0];
}

// The same left-to-right fp32 sum, bit for bit, without walking V dependent adds.
//
// While the running sum s stays inside one binade [2^k, 2^(k+1)) it is an integer multiple m*q of
// q = 2^(k-23), and fl(s + a) = (m + r)*q where r = round-to-nearest-even of a/q -- the tie (a/q = n + 1/2)
// goes by the parity of m + n, nothing else about m matters.  So a run of elements inside one binade
// advances m by an integer that depends only on the parity of the incoming m: every thread computes both
// candidates (inc[0], inc[1]) for its contiguous run in parallel, using a binade guessed from an
// ordinary block scan, and one lane then chains the 1024 runs with integer adds, checking each guess
// exactly (incoming exponent == k, outgoing m <= 2^24).  Runs that cross a binade, start at 0 or sit too
// close to a power of two to be guessed are "slow": their elements are parked in LDS and the lane adds
// them one by one in fp32.  A distribution crosses ~25 binades on its way to 1, so a few dozen runs
// are slow and the rest cost one chain step each.  A chain step is ~45 scalar instructions at a single wave's issue
// rate (one per ~5 cycles): 1024 of them took ~120 us, so each wave first composes its 64 runs (phase C0) and the
// chain is per 64-run group wherever a group is uniform.  Measured per sampled token (one workgroup): 0.15 ms at
// V = 32000, 0.34 ms at V = 152064; the one-lane walk: 0.45 / 2.6 ms.
constexpr int kSlowSlots = 56;          // LDS slots for slow runs (more fall back to global reads)
constexpr int kMaxRun = 152;            // elements per thread: V <= 1024 * 152 = 155648 (Qwen2: 152064)
struct OrderedSumLds {
    float s_in[1025];                   // exact running sum entering each run (s_in[1024] = total)
    int inc0[1024], inc1[1024];         // mantissa increment of a fast run for even / odd incoming mantissa
    short kexp[1024];                   // guessed biased exponent of a fast run, -1: slow
    short slot[1024];                   // LDS slot of a slow run, -1: read from global
    float wtot[16];
    int g_mode[16], g_ke[16], g_inc0[16], g_inc1[16];   // per 64-run group: 0 chain it, 1 uniform (one step), 2 empty
    uint32_t g_sb[16];                    // bits of the sum entering a uniform / empty group
    int nslots;
    float parked[kSlowSlots * kMaxRun];
};

// a slow run, one fp32 add at a time (wave-uniform: every lane computes the same chain)
__device__ inline uint32_t walk_run(uint32_t sbits, const float *__restrict__ p, int c0, int c1, int slot, const float *parked) {
    float s = __uint_as_float(sbits);
    if (slot >= 0) {
        const float *q = parked + slot * kMaxRun;                 // 16-byte aligned: kMaxRun % 4 == 0
        const int n = c1 - c0;
        int j = 0;
        for (; j + 8 <= n; j += 8) {                              // two b128 reads per 8 dependent adds
            const float4v a = *reinterpret_cast<const float4v *>(q + j), b = *reinterpret_cast<const float4v *>(q + j + 4);
            s = __fadd_rn(s, a[0]); s = __fadd_rn(s, a[1]); s = __fadd_rn(s, a[2]); s = __fadd_rn(s, a[3]);
            s = __fadd_rn(s, b[0]); s = __fadd_rn(s, b[1]); s = __fadd_rn(s, b[2]); s = __fadd_rn(s, b[3]);
        }
        for (; j < n; j++) s = __fadd_rn(s, q[j]);
    }
    else for (int i = c0; i < c1; i++) s = __fadd_rn(s, p[i]);
    return __float_as_uint(s);
}

__device__ inline float ordered_sum(const float *__restrict__ p, int V, OrderedSumLds &L) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int run = ((V + 1023) / 1024 + 3) & ~3;
    const int i0 = min(V, tid * run), i1 = min(V, i0 + run), n = i1 - i0;
    const bool vec = (V & 3) == 0;
    // phase A: estimated prefix (any order will do: it only picks the binade to try)
    float cs = 0.f;
    if (vec) for (int i = i0; i < i1; i += 4) { const float4v v = *reinterpret_cast<const float4v *>(p + i); cs += (v[0] + v[1]) + (v[2] + v[3]); }
    else for (int i = i0; i < i1; i++) cs += p[i];
    float inc = cs;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float up = __shfl_up(inc, o, 64); if (lane >= o) inc += up; }
    if (tid == 0) L.nslots = 0;
    __syncthreads();
    if (lane == 63) L.wtot[wave] = inc;
    __syncthreads();
    float wbase = 0.f;
    for (int w = 0; w < wave; w++) wbase += L.wtot[w];
    const float est_out = wbase + inc, est_in = est_out - cs;
    // phase B: classify the run and, if it looks like it stays in one binade, take both increments
    const uint32_t eb = __float_as_uint(est_in) >> 23;              // biased exponent (est_in >= 0)
    const float lo = __uint_as_float(eb << 23), hi = __uint_as_float((eb + 1) << 23);
    const bool fast = n > 0 && eb > 24 && eb < 250 && est_in >= lo * 1.001f && est_out <= hi * 0.999f;
    short myslot = -1;
    if (fast) {
        const float inv_q = __uint_as_float((uint32_t)(127 + 150 - (int)eb) << 23);      // 2^(23 - k), k = eb - 127
        int a0 = 0, a1 = 1;
        auto step = [&](float a) {
            const float t = a * inv_q;                              // exact (power of two), < 2^24 in a one-binade run
            const float fl = floorf(t);
            const float fr = t - fl;                                // exact
            const int nn = (int)fl + (fr > 0.5f ? 1 : 0);
            const int tie = fr == 0.5f ? 1 : 0;
            a0 += nn + (tie & (a0 + (int)fl));
            a1 += nn + (tie & (a1 + (int)fl));
        };
        if (vec) for (int i = i0; i < i1; i += 4) { const float4v v = *reinterpret_cast<const float4v *>(p + i); step(v[0]); step(v[1]); step(v[2]); step(v[3]); }
        else for (int i = i0; i < i1; i++) step(p[i]);
        L.inc0[tid] = a0; L.inc1[tid] = a1 - 1;
        L.kexp[tid] = (short)eb;
    } else {
        L.kexp[tid] = -1;
        if (n > 0 && n <= kMaxRun) {
            const int sl = atomicAdd(&L.nslots, 1);
            if (sl < kSlowSlots) { myslot = (short)sl; for (int i = i0; i < i1; i++) L.parked[sl * kMaxRun + (i - i0)] = p[i]; }
        }
    }
    L.slot[tid] = myslot;
    __syncthreads();
    // phase C0 (all 16 waves): a wave composes its 64 runs.  The fast step m -> m + inc[m & 1] acts on the mantissa
    // parity only through inc, so two consecutive steps compose to another such pair,
    //     (L then R)[p] = L[p] + R[(p + L[p]) & 1],
    // which is associative: a wave-level scan gives every run its increment from the start of the group for either
    // entering parity, and the group's total.  A group is "uniform" when all its runs are fast in the same binade.
    const int c_me = tid, ke_me = L.kexp[c_me];
    const bool empty_me = n == 0;
    int t0 = empty_me ? 0 : L.inc0[c_me], t1 = empty_me ? 0 : L.inc1[c_me];        // inclusive transducer prefix (so far: own)
    {
        const unsigned long long fastm = __ballot(!empty_me && ke_me >= 0), usedm = __ballot(!empty_me);
        const int first = usedm ? __builtin_ctzll(usedm) : 0;
        const int ke_ref = __shfl(ke_me, first, 64);
        const bool uniform = usedm != 0 && fastm == usedm && __ballot(!empty_me && ke_me != ke_ref) == 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int l0 = __shfl_up(t0, o, 64), l1 = __shfl_up(t1, o, 64);
            if (lane >= o) {
                const int n0v = l0 + ((l0 & 1) ? t1 : t0), n1v = l1 + (((1 + l1) & 1) ? t1 : t0);
                t0 = n0v; t1 = n1v;
            }
        }
        if (lane == 63) { L.g_mode[wave] = usedm == 0 ? 2 : (uniform ? 1 : 0); L.g_ke[wave] = ke_ref; L.g_inc0[wave] = t0; L.g_inc1[wave] = t1; }
    }
    __syncthreads();
    // phase C1: wave 0 chains the 16 groups.  A uniform group is one integer step (checked exactly: entering
    // exponent == its binade, leaving mantissa <= 2^24); any other group is chained run by run -- per-run data in vector
    // registers (lane j = run 64g + j) read back with v_readlane, slow runs walked in fp32.
    if (wave == 0) {
        uint32_t sb = 0;                                             // bits of the running sum (>= 0)
        for (int g = 0; g < 16; g++) {
            const int mode = L.g_mode[g];
            if (mode == 2) { if (lane == 0) L.g_sb[g] = sb; continue; }
            if (mode == 1) {
                const int ke = L.g_ke[g];
                const uint32_t m = (sb & 0x7fffffu) | 0x800000u;
                const uint32_t m2 = m + (uint32_t)((m & 1) ? L.g_inc1[g] : L.g_inc0[g]);
                if ((int)(sb >> 23) == ke && m2 <= 0x1000000u) {
                    if (lane == 0) L.g_sb[g] = sb;
                    sb = ((uint32_t)ke << 23) + (m2 - 0x800000u);
                    continue;
                }
                if (lane == 0) L.g_mode[g] = 0;                      // the guess did not hold: chain it
            }
            const int c = g * 64 + lane;
            const int ke_v = L.kexp[c], a0_v = L.inc0[c], a1_v = L.inc1[c], sl_v = L.slot[c];
            uint32_t sin_v = 0;
#pragma unroll 1
            for (int j = 0; j < 64; j++) {
                if (lane == j) sin_v = sb;
                const int c0 = min(V, (g * 64 + j) * run), c1 = min(V, c0 + run);
                const int ke = __builtin_amdgcn_readlane(ke_v, j);
                const int a0 = __builtin_amdgcn_readlane(a0_v, j), a1 = __builtin_amdgcn_readlane(a1_v, j);
                // branch-free fast step: the new bits are (ke << 23) + (m2 - 2^23), also right when m2 == 2^24
                const uint32_t m = (sb & 0x7fffffu) | 0x800000u;
                const uint32_t m2 = m + (uint32_t)(a0 + (int)(m & 1) * (a1 - a0));
                const bool ok = (ke >= 0) & ((int)(sb >> 23) == ke) & (m2 <= 0x1000000u);
                const uint32_t fast_sb = ((uint32_t)ke << 23) + (m2 - 0x800000u);
                if (!ok && c0 < c1) sb = __builtin_amdgcn_readfirstlane(walk_run(sb, p, c0, c1, __builtin_amdgcn_readlane(sl_v, j), L.parked));
                else sb = (ok & (c0 < c1)) ? fast_sb : sb;
            }
            L.s_in[c] = __uint_as_float(sin_v);
        }
        if (lane == 0) L.s_in[1024] = __uint_as_float(sb);
    }
    __syncthreads();
    // phase C2 (all waves): runs of a uniform group get their entering sum from the group's and their scan prefix
    {
        const int mode = L.g_mode[wave];
        if (mode != 0) {
            const uint32_t gsb = L.g_sb[wave];
            uint32_t mine = gsb;
            if (mode == 1) {
                const int e0 = __shfl_up(t0, 1, 64), e1 = __shfl_up(t1, 1, 64);        // exclusive prefix = inclusive of lane - 1
                const uint32_t m = (gsb & 0x7fffffu) | 0x800000u;
                const int excl = lane == 0 ? 0 : ((m & 1) ? e1 : e0);
                mine = ((uint32_t)L.g_ke[wave] << 23) + (m + (uint32_t)excl - 0x800000u);
            }
            L.s_in[tid] = __uint_as_float(mine);
        }
    }
    __syncthreads();
    return L.s_in[1024];
}

// Sampling::All { temperature } (candle-transformers LogitsProcessor over rand 0.8's WeightedIndex<f32>):
//   prs = softmax(logits * (f32)(1/temperature));  total = sum(prs);  chosen = uniform[0,1) * total with
//   uniform = f32::from_bits((next_u32() >> 9) | 0x3f800000) - 1;  token = #{ j < V-1 : prs[0]+..+prs[j] <= chosen }.
// Both sums of the reference (softmax denominator: candle's scalar vec_sum in a default build; cumulative
// weights: WeightedIndex::new) run left to right in fp32, and at V ~ 1e5 that order is visible in the
// result (small terms are absorbed: the sequential total is off by ~4e-5, several tokens wide in a flat
// region), so the two sums are done in that order by one lane; max / exp / divide / count are parallel.
// exp is evaluated in fp64 and rounded, which reproduces a correctly rounded expf (glibc's, which Rust's
// f32::exp calls) except in ~1e-9 of the cases.
__device__ inline int sample_all(const float *__restrict__ logits, int V, SampleState *__restrict__ ss, float *__restrict__ p,
                                 float *red, unsigned char *lds, float *bcast, int *count) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float mul = ss->inv_temp;
    // on == 1: ordered_sum (parallel, bit-identical to the walk); on == 2 or a vocabulary beyond its run
    // size: sequential_sum (the plain walk, kept as the cross-check)
    const bool walk = ss->on == 2 || V > 1024 * kMaxRun;
    OrderedSumLds &L = *reinterpret_cast<OrderedSumLds *>(lds);
    float *tiles = reinterpret_cast<float *>(lds);
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, __fmul_rn(logits[i], mul));
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = red[0];
    for (int w = 1; w < 16; w++) mx = fmaxf(mx, red[w]);
    for (int i = tid; i < V; i += 1024) p[i] = (float)exp((double)__fsub_rn(__fmul_rn(logits[i], mul), mx));
    __syncthreads();
    const float S = walk ? sequential_sum(p, V, tiles, bcast, false) : ordered_sum(p, V, L);
    for (int i = tid; i < V; i += 1024) p[i] = __fdiv_rn(p[i], S);
    if (tid == 0) *count = 0;
    __syncthreads();
    // walk: p[] is replaced by the cumulative weights; ordered: L.s_in[] holds the sum entering every run
    const float total = walk ? sequential_sum(p, V, tiles, bcast, true) : ordered_sum(p, V, L);
    if (tid == 0) {
        // UniformFloat::new(0, total): scale = total, lowered by ulps while scale * (1 - 2^-23) >= total
        float scale = total;
        while (__fmul_rn(scale, 1.0f - 1.1920929e-07f) >= total) scale = __uint_as_float(__float_as_uint(scale) - 1);
        const uint64_t w = ((uint64_t)ss->draw_hi << 32) | ss->draw_lo;
        const uint32_t u = chacha12_word(ss->key, w);
        ss->draw_lo = (uint32_t)(w + 1); ss->draw_hi = (uint32_t)((w + 1) >> 32);
        bcast[1] = __fmul_rn(__fsub_rn(__uint_as_float((u >> 9) | 0x3f800000u), 1.0f), scale);
    }
    __syncthreads();
    const float chosen = bcast[1];
    // partition_point: how many cumulative weights are <= chosen (chosen < total, so the last one never is)
    int n = 0;
    if (walk) {
        for (int i = tid; i < V - 1; i += 1024) n += p[i] <= chosen ? 1 : 0;
    } else {
        const int run = ((V + 1023) / 1024 + 3) & ~3;
        const int i0 = min(V, tid * run), i1 = min(V, i0 + run);
        if (i0 < i1) {
            if (L.s_in[tid + 1] <= chosen) n = i1 - i0;            // the whole run is below (sums only grow)
            else if (L.s_in[tid] <= chosen) {                       // the boundary run: walk it from its exact start
                float cum = L.s_in[tid];
                for (int i = i0; i < i1; i++) { cum = __fadd_rn(cum, p[i]); if (cum <= chosen) n++; else break; }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (lane == 0 && n) atomicAdd(count, n);
    __syncthreads();
    return min(*count, V - 1);
}

__device__ __forceinline__ void select_advance_body(const float *__restrict__ logits, int V, StepState *__restrict__ st,
                                           SampleState *__restrict__ ss, float *__restrict__ scratch,
                                           uint32_t *__restrict__ out_tokens, int advance, unsigned char *lds,
                                           const ArgmaxCand *__restrict__ cand = nullptr) {
    __shared__ float bv[16], bcast[2];
    __shared__ int bi[16], count;
    const int tid = threadIdx.x;
    int idx;
    if (ss->on) idx = sample_all(logits, V, ss, scratch, bv, lds, bcast, &count);
    else if (cand) idx = argmax_candidates(cand, bv, bi);
    else idx = argmax_last(logits, V, bv, bi);
    if (tid == 0) {
        const uint32_t tok = (uint32_t)idx;
        if (out_tokens) out_tokens[st->step] = tok;
        st->token = tok;
        if (st->eos >= 0 && tok == (uint32_t)st->eos) st->done = 1;
        if (advance) { st->pos += 1; st->len += 1; st->step += 1; }
    }
}

constexpr size_t kSelLds = sizeof(OrderedSumLds) > 2 * kSelTile * 4 ? sizeof(OrderedSumLds) : 2 * kSelTile * 4;

__global__ __launch_bounds__(1024) void select_advance_kernel(const float *__restrict__ logits, int V,
                                                              StepState *__restrict__ st, SampleState *__restrict__ ss,
                                                              float *__restrict__ scratch, uint32_t *__restrict__ out_tokens,
                                                              int advance, const ArgmaxCand *__restrict__ cand, uint32_t *epoch_bump) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kSelLds];
    // the persistent decode engine's tag epoch (k_engine.hip): one step per forward, so no two launches ever share a tag
    if (epoch_bump && threadIdx.x == 0) *epoch_bump += 1;
    select_advance_body(logits, V, st, ss, scratch, out_tokens, advance, lds, cand);
}

// one workgroup per sequence of a batch: logits [B][V], state and scratch from the SeqRef
__global__ __launch_bounds__(1024) void select_advance_batch_kernel(const float *__restrict__ logits, int V,
                                                                    const SeqRef *__restrict__ seqs, int advance) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[kSelLds];
    const SeqRef sq = seqs[blockIdx.x];
    select_advance_body(logits + (size_t)blockIdx.x * V, V, sq.st, sq.ss, sq.sel_scratch, sq.out_tokens, advance, lds);
}

int launch_select_advance_batch(Launcher &L, const float *logits, int64_t V, const SeqRef *seqs_dev, int B, int advance) {
    return L.launch(KC_ARGMAX, (double)V * 4 * B, 0, select_advance_batch_kernel, dim3((unsigned)B), dim3(1024), 0, logits, (int)V,
                    seqs_dev, advance);
}

// a tensor-parallel batch's gathered logits [tp][B][Vs] (rank-major: what one all-gather of every rank's [B][Vs] block leaves) ->
// [B][V] (V = tp * Vs), the layout token selection and fl_batch_forward's caller read
__global__ __launch_bounds__(256) void unshard_logits_kernel(const float *__restrict__ in, float *__restrict__ out, int tp, int B, int Vs) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4, n = (long long)tp * B * Vs;
    if (i >= n) return;
    const int r = (int)(i / ((long long)B * Vs)), rem = (int)(i % ((long long)B * Vs)), b = rem / Vs, j = rem % Vs;     // (Vs % 4 == 0: a float4 stays in its row)
    *reinterpret_cast<float4v *>(out + ((size_t)b * tp + r) * Vs + j) = *reinterpret_cast<const float4v *>(in + i);
}
int launch_unshard_logits(Launcher &L, const float *in, float *out, int tp, int B, int64_t Vs) {
    if (Vs % 4) FL_FAIL(FL_ERR_UNSUPPORTED, "a tensor-parallel batch needs a vocabulary shard that is a multiple of 4");
    const long long n = (long long)tp * B * Vs;
    return L.launch(KC_ARGMAX, (double)n * 8, 0, unshard_logits_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, in, out, tp, B, (int)Vs);
}

// scratch: V floats (probabilities / cumulative weights of the sampling path)
int launch_select_advance(Launcher &L, const float *logits, int64_t V, StepState *st, SampleState *ss, float *scratch,
                          uint32_t *out_tokens, int advance, const ArgmaxCand *cand, uint32_t *epoch_bump) {
    return L.launch(KC_ARGMAX, (double)V * 4, 0, select_advance_kernel, dim3(1), dim3(1024), 0, logits, (int)V, st, ss,
                    scratch, out_tokens, advance, cand, epoch_bump);
}

// ------------------------------------------------------------------------------- local shard reduce
// FL_TP_EMULATED: all shards live on one GPU, so all-reduce(sum) is a local sum written back to
// every shard's buffer.  Fixed summation order s = 0..n-1 (reproducible).
__global__ __launch_bounds__(256) void reduce_shards_kernel(float *const *__restrict__ bufs, int nshards, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < nshards; k++) s += bufs[k][i];
    for (int k = 0; k < nshards; k++) bufs[k][i] = s;
}

int launch_reduce_shards(Launcher &L, float *const *bufs_dev, int nshards, int64_t n) {
    return L.launch(KC_REDUCE, (double)n * 8 * nshards, 0, reduce_shards_kernel, dim3((unsigned)((n + 255) / 256)),
                    dim3(256), 0, bufs_dev, nshards, (long long)n);
}

// ------------------------------------------------------------------------------- weight conversion
__device__ inline float load_as_f32(const void *p, size_t i, int dt) {
    if (dt == FL_DTYPE_F32) return reinterpret_cast<const float *>(p)[i];
    if (dt == FL_DTYPE_BF16) return bf16_bits_to_float(reinterpret_cast<const bf16_t *>(p)[i]);
    return (float)reinterpret_cast<const _Float16 *>(p)[i];
}

template <typename DT>
__global__ __launch_bounds__(256) void convert_slice_kernel(const void *__restrict__ src, int src_dt, long long src_ld,
                                                            long long r0, long long c0, long long rows, long long cols,
                                                            DT *__restrict__ dst, long long dst_ld, long long dst_row0,
                                                            int row_mode, int head_pad, int dm, int d) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols) return;
    const long long r = i / cols, c = i % cols;
    const float v = load_as_f32(src, (size_t)((r0 + r) * src_ld + c0 + c), src_dt);
    long long dr = row_mode == 0 ? dst_row0 + r : gateup_row(r, row_mode == 2), dc = c;
    // head_pad: element j of a head of the model's head_dim dm goes to j (first half) or d/2 + j - dm/2 (second half) of a padded
    // head of d: rotate-half partners stay d/2 apart; 1 = the slice's rows are heads, 2 = its columns are
    auto padded = [&](long long x) { const long long hd = x / dm, j = x % dm; return hd * d + (j < dm / 2 ? j : d / 2 + j - dm / 2); };
    if (head_pad == 1) dr = dst_row0 + padded(r);
    if (head_pad == 2) dc = padded(c);
    elem<DT>::st(dst + (size_t)(dr * dst_ld + dc), v);
}

int launch_convert_slice(Launcher &L, int src_dtype, const void *src, int64_t src_ld, int64_t r0, int64_t c0,
                         int64_t rows, int64_t cols, int dst_dtype, void *dst, int64_t dst_ld, int64_t dst_row0,
                         int row_mode, int head_pad, int64_t dm, int64_t d) {
    const int64_t total = rows * cols;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    if (dst_dtype == FL_DTYPE_BF16)
        return L.launch(KC_CONVERT, 0, 0, convert_slice_kernel<bf16_t>, dim3(blocks), dim3(256), 0, src, src_dtype,
                        (long long)src_ld, (long long)r0, (long long)c0, (long long)rows, (long long)cols,
                        (bf16_t *)dst, (long long)dst_ld, (long long)dst_row0, row_mode, head_pad, (int)dm, (int)d);
    return L.launch(KC_CONVERT, 0, 0, convert_slice_kernel<float>, dim3(blocks), dim3(256), 0, src, src_dtype,
                    (long long)src_ld, (long long)r0, (long long)c0, (long long)rows, (long long)cols, (float *)dst,
                    (long long)dst_ld, (long long)dst_row0, row_mode, head_pad, (int)dm, (int)d);
}

int launch_convert_vec_f32(Launcher &L, int src_dtype, const void *src, int64_t off, int64_t n, float *dst) {
    return launch_convert_slice(L, src_dtype, src, n + off, 0, off, 1, n, FL_DTYPE_F32, dst, n, 0, 0);
}

}  // namespace fl
