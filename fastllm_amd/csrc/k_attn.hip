// k_attn.hip -- single-sequence KV-cached attention.
//
// Cache layout per layer: K and V as [Hkv][max_seq][d] (rotated K, App. A.4/A.8).  GQA is done
// by indexing kv_head = q_head / (H/Hkv) (App. A.6) -- K/V are never replicated, and one
// workgroup serves all G = H/Hkv query heads of its kv head so each K/V byte is read once.
//
// Key rows are read straight to VGPRs, 16 B per lane: d/8 lanes cover one key row, so a wave
// instruction fetches 64/(d/8) keys.  Each (wave, key-slot) keeps its own online-softmax state
// (running max m, sum l, output o) -- no cross-lane traffic in the loop except the d/8-lane
// dot-product reduce (wavefront shuffles) -- and the states are merged once at the end:
// key-slots by shuffles, waves through LDS, split-S workgroups through a small fp32 partial
// buffer + combine kernel.
#include <stdlib.h>

#include <algorithm>

#include "attn_common.h"

namespace fl {

template <int GMAX>
struct AttnState {
    float m[GMAX], l[GMAX], o[GMAX][8];
    __device__ void init() {
#pragma unroll
        for (int g = 0; g < GMAX; g++) {
            m[g] = -INFINITY; l[g] = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) o[g][j] = 0.f;
        }
    }
};

__device__ inline void merge_state(float &m, float &l, float (&o)[8], float m2, float l2, const float (&o2)[8]) {
    const float M = fmaxf(m, m2);
    if (M == -INFINITY) return;                       // both empty
    const float a = __expf(m - M), b = __expf(m2 - M);
    l = l * a + l2 * b;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = o[j] * a + o2[j] * b;
    m = M;
}

// Accumulate keys [lo, hi) of one kv head into this lane's state.  Uniform trip count per wave.
// Two key batches per iteration: all four 16-B loads are issued before the first dot product.
template <typename CT, int D, int GMAX>
__device__ inline void attend_one(AttnState<GMAX> &s, const float (&q)[GMAX][8], int G, const float (&kv)[8],
                                  const float (&vv)[8], bool valid) {
    constexpr int LPK = D / 8;
#pragma unroll
    for (int g = 0; g < GMAX; g++) {
        if (g < G) {
            float dot = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j++) dot = fmaf(q[g][j], kv[j], dot);
#pragma unroll
            for (int off = 1; off < LPK; off <<= 1) dot += __shfl_xor(dot, off, 64);
            if (valid) {
                const float mn = fmaxf(s.m[g], dot);
                const float alpha = __expf(s.m[g] - mn), p = __expf(dot - mn);
                s.l[g] = s.l[g] * alpha + p;
#pragma unroll
                for (int j = 0; j < 8; j++) s.o[g][j] = s.o[g][j] * alpha + p * vv[j];
                s.m[g] = mn;
            }
        }
    }
}

// NW waves share the key range; UNR key batches are loaded (K and V, 16 B per lane each) before
// the first dot product so 2*UNR wave-loads are in flight per wave.
template <typename CT, int D, int GMAX, int NW, int UNR>
__device__ inline void attend_range(AttnState<GMAX> &s, const float (&q)[GMAX][8], int G, const CT *__restrict__ kc,
                                    const CT *__restrict__ vc, int lo, int hi, int wave, int lane) {
    constexpr int LPK = D / 8, KPI = 64 / LPK;
    const int li = lane % LPK, ks = lane / LPK;
    for (int base = lo + wave * KPI; base < hi; base += UNR * NW * KPI) {
        float kf[UNR][8], vf[UNR][8];
        bool valid[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) {
            const int key = base + u * NW * KPI + ks;
            valid[u] = key < hi;
            const int kk = valid[u] ? key : hi - 1;
            load8(kc + (size_t)kk * D + li * 8, kf[u]);
            load8(vc + (size_t)kk * D + li * 8, vf[u]);
        }
#pragma unroll
        for (int u = 0; u < UNR; u++) attend_one<CT, D, GMAX>(s, q, G, kf[u], vf[u], valid[u]);
    }
}

// Merge the per-lane states of a workgroup.  First the KPI key-slot states of each wave (xor
// shuffles), then the NW waves through LDS: lds [NW][GMAX][D + 2] floats holds (o[D], m, l) per
// (wave, head).  After the barrier every thread can combine any (head, d-slice) from LDS.
template <int D, int GMAX, int NW>
__device__ inline void merge_to_lds(AttnState<GMAX> &s, int G, float *lds, int wave, int lane) {
    constexpr int LPK = D / 8, STR = D + 2;
    const int li = lane % LPK;
#pragma unroll
    for (int g = 0; g < GMAX; g++) {
        if (g < G) {
#pragma unroll
            for (int off = LPK; off < 64; off <<= 1) {
                float m2 = __shfl_xor(s.m[g], off, 64), l2 = __shfl_xor(s.l[g], off, 64), o2[8];
#pragma unroll
                for (int j = 0; j < 8; j++) o2[j] = __shfl_xor(s.o[g][j], off, 64);
                merge_state(s.m[g], s.l[g], s.o[g], m2, l2, o2);
            }
        }
    }
    if (lane < LPK) {
#pragma unroll
        for (int g = 0; g < GMAX; g++) {
            if (g < G) {
                float *p = lds + ((size_t)wave * GMAX + g) * STR;
#pragma unroll
                for (int j = 0; j < 8; j++) p[li * 8 + j] = s.o[g][j];
                if (li == 0) { p[D] = s.m[g]; p[D + 1] = s.l[g]; }
            }
        }
    }
    __syncthreads();
}

template <typename CT, int D, int GMAX>
__device__ inline void load_q(float (&q)[GMAX][8], const CT *__restrict__ qrow, int hq0, int G, int lane, float scale) {
    constexpr int LPK = D / 8;
    const int li = lane % LPK;
#pragma unroll
    for (int g = 0; g < GMAX; g++) {
        if (g < G) {
            load8(qrow + (size_t)(hq0 + g) * D + li * 8, q[g]);
#pragma unroll
            for (int j = 0; j < 8; j++) q[g][j] *= scale;       // (q.k) * 1/sqrt(d), App. A.3
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) q[g][j] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------- decode (T = 1)
// grid (Hkv, nsplit, ceil(G/GMAX)), NW waves per workgroup.  T == 1: no mask, no window -- the
// whole cache is visible.  A decode step is a chain of short dependent kernels, so the kernel is
// built to minimise serial memory round trips rather than to spread over many CUs: with
// nsplit == 1 (cache capacity up to 2048 positions per split) ONE 16-wave workgroup per kv head
// streams that head's K/V (<= ~1 MB) and writes the normalised output directly -- no partial
// buffers, no fences, no second launch.  Longer caches split S across workgroups; their (m, l, o)
// slabs are merged in the same launch by the workgroup that draws the last ticket (plain stores ->
// per-wave vmcnt(0) -> barrier -> agent-scope release -> vmcnt(0) -> ticket; last arriver: acquire
// -> barrier -> plain loads; cdna guide Guideline 16 counter form, placement-independent).  The
// ticket word is reset by the last arriver, so a captured graph replays without a memset node.
template <typename CT, int D, int GMAX, int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(const CT *__restrict__ q, const CT *__restrict__ kc,
                                                              const CT *__restrict__ vc, const StepState *__restrict__ st,
                                                              float *__restrict__ part_m, float *__restrict__ part_l,
                                                              float *__restrict__ part_o, unsigned *__restrict__ counters,
                                                              CT *__restrict__ out, int H, int Hkv, int max_seq,
                                                              float scale, int nsplit) {
    __shared__ float lds[decode_lds_floats<D, GMAX, NW>()];
    __shared__ int is_last;
    constexpr int LPK = D / 8, KPI = 64 / LPK, UNR = 2;
    const int hk = blockIdx.x, split = blockIdx.y;
    const int Gall = H / Hkv, g0 = blockIdx.z * GMAX;
    const int G = min(GMAX, Gall - g0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = (int)st->len + 1;
    int per = (S + nsplit - 1) / nsplit;
    per = (per + NW * KPI - 1) / (NW * KPI) * (NW * KPI);
    const int lo = split * per, hi = min(S, lo + per);
    const int hq0 = hk * Gall + g0;

    float qv[GMAX][8];
    load_q<CT, D, GMAX>(qv, q, hq0, G, lane, scale);
    AttnState<GMAX> s; s.init();
    if (lo < hi)
        attend_range<CT, D, GMAX, NW, UNR>(s, qv, G, kc + (size_t)hk * max_seq * D, vc + (size_t)hk * max_seq * D, lo, hi, wave, lane);
    merge_to_lds<D, GMAX, NW>(s, G, lds, wave, lane);

    decode_tail<CT, D, GMAX, NW>(lds, &is_last, G, hq0, hk * (int)gridDim.z + (int)blockIdx.z, split, nsplit, part_m, part_l,
                                 part_o, counters, out);
}

template <typename CT, int D, int GMAX, int NW>
static int launch_decode_t(Launcher &L, const void *q, const void *kc, const void *vc, const StepState *st,
                           void *out, const AttnScratch &sc, int64_t H, int64_t Hkv, int64_t max_seq, float scale) {
    const int G = (int)(H / Hkv);
    if (sc.nsplit > 64) FL_FAIL(FL_ERR_BAD_ARGUMENT, "attention: at most 64 splits");
    dim3 grid((unsigned)Hkv, (unsigned)sc.nsplit, (unsigned)((G + GMAX - 1) / GMAX));
    // the KV length lives on the device; the caller passes its host copy for the byte accounting
    double kvbytes = 2.0 * (double)sc.kv_len_hint * Hkv * D * sizeof(CT);
    return L.launch(KC_ATTN_DECODE, kvbytes, 4.0 * (double)sc.kv_len_hint * H * D, attn_decode_kernel<CT, D, GMAX, NW>, grid,
                    dim3(NW * 64), 0, (const CT *)q, (const CT *)kc, (const CT *)vc, st, sc.part_m, sc.part_l, sc.part_o,
                    sc.counters, (CT *)out, (int)H, (int)Hkv, (int)max_seq, scale, sc.nsplit);
}

// The same kernel body for the rows of a decode batch (caches in the plain layout: fp32 models, head shapes outside the MFMA
// attention): grid.y = sequence x split, every pointer, the length and the split count from the sequence's SeqRef.
template <typename CT, int D, int GMAX, int NW>
__global__ __launch_bounds__(NW * 64) void attn_decode_batch_kernel(const CT *__restrict__ q, const SeqRef *__restrict__ seqs, size_t kv_layer_off,
                                                                    CT *__restrict__ out, int H, int Hkv, float scale, int max_nsplit) {
    __shared__ float lds[decode_lds_floats<D, GMAX, NW>()];
    __shared__ int is_last;
    constexpr int LPK = D / 8, KPI = 64 / LPK, UNR = 2;
    const int hk = blockIdx.x, b = blockIdx.y / max_nsplit, split = blockIdx.y % max_nsplit;
    const SeqRef &sq = seqs[b];
    const int nsplit = sq.nsplit;
    if (split >= nsplit) return;                              // (the whole workgroup: no barrier is left waiting)
    const int Gall = H / Hkv, g0 = blockIdx.z * GMAX;
    const int G = min(GMAX, Gall - g0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = (int)sq.st->len + 1;
    int per = (S + nsplit - 1) / nsplit;
    per = (per + NW * KPI - 1) / (NW * KPI) * (NW * KPI);
    const int lo = split * per, hi = min(S, lo + per);
    const int hq0 = hk * Gall + g0;
    const size_t sa = (size_t)sq.seq_alloc;
    const CT *kc = reinterpret_cast<const CT *>(sq.k) + kv_layer_off * sa, *vc = reinterpret_cast<const CT *>(sq.v) + kv_layer_off * sa;

    float qv[GMAX][8];
    load_q<CT, D, GMAX>(qv, q + (size_t)b * H * D, hq0, G, lane, scale);
    AttnState<GMAX> s; s.init();
    if (lo < hi)
        attend_range<CT, D, GMAX, NW, UNR>(s, qv, G, kc + (size_t)hk * sa * D, vc + (size_t)hk * sa * D, lo, hi, wave, lane);
    merge_to_lds<D, GMAX, NW>(s, G, lds, wave, lane);
    decode_tail<CT, D, GMAX, NW>(lds, &is_last, G, hq0, hk * (int)gridDim.z + (int)blockIdx.z, split, nsplit, sq.part_m, sq.part_l,
                                 sq.part_o, sq.counters, out + (size_t)b * H * D);
}

template <typename CT, int D, int GMAX, int NW>
static int launch_decode_batch_t(Launcher &L, const void *q, const SeqRef *seqs, int B, int max_nsplit, size_t kv_layer_off, void *out,
                                 int64_t H, int64_t Hkv, float scale) {
    const int G = (int)(H / Hkv);
    if (max_nsplit > 64 || (int64_t)B * max_nsplit > 65535) FL_FAIL(FL_ERR_BAD_ARGUMENT, "attention: at most 64 splits, 65535 sequence x split workgroup rows");
    dim3 grid((unsigned)Hkv, (unsigned)(B * max_nsplit), (unsigned)((G + GMAX - 1) / GMAX));
    return L.launch(KC_ATTN_DECODE, 0.0, 0.0, attn_decode_batch_kernel<CT, D, GMAX, NW>, grid, dim3(NW * 64), 0, (const CT *)q, seqs, kv_layer_off,
                    (CT *)out, (int)H, (int)Hkv, scale, max_nsplit);
}

bool attn_decode_batch_supported(int64_t d) { return d == 64 || d == 128; }

int launch_attn_decode_batch(Launcher &L, int dtype, const void *q, const SeqRef *seqs_dev, int B, int max_nsplit, size_t kv_layer_off, void *out,
                             int64_t H, int64_t Hkv, int64_t d, float scale) {
    const bool small = H / Hkv <= 4;
#define FL_DISPATCH(CT)                                                                                                                   \
    if (d == 128) return small ? launch_decode_batch_t<CT, 128, 4, 4>(L, q, seqs_dev, B, max_nsplit, kv_layer_off, out, H, Hkv, scale)    \
                               : launch_decode_batch_t<CT, 128, 8, 4>(L, q, seqs_dev, B, max_nsplit, kv_layer_off, out, H, Hkv, scale);   \
    if (d == 64) return small ? launch_decode_batch_t<CT, 64, 4, 4>(L, q, seqs_dev, B, max_nsplit, kv_layer_off, out, H, Hkv, scale)      \
                              : launch_decode_batch_t<CT, 64, 8, 4>(L, q, seqs_dev, B, max_nsplit, kv_layer_off, out, H, Hkv, scale);
    if (dtype == FL_DTYPE_BF16) { FL_DISPATCH(bf16_t) }
    else { FL_DISPATCH(float) }
#undef FL_DISPATCH
    FL_FAIL(FL_ERR_UNSUPPORTED, "attention: head_dim %lld not supported (64 or 128)", (long long)d);
}

int launch_attn_decode(Launcher &L, int dtype, const void *q, const void *k_cache, const void *v_cache,
                       const StepState *st, void *out, const AttnScratch &sc, int64_t H, int64_t Hkv, int64_t d,
                       int64_t max_seq, float scale) {
    const int G = (int)(H / Hkv);
    const bool small = G <= 4;
    const int nw = tune(TK_ATTN_NW);
#define FL_DISPATCH(CT)                                                                                        \
    if (d == 128 && nw >= 16) return small ? launch_decode_t<CT, 128, 4, 16>(L, q, k_cache, v_cache, st, out, sc, H, Hkv, max_seq, scale) \
                               : launch_decode_t<CT, 128, 8, 8>(L, q, k_cache, v_cache, st, out, sc, H, Hkv, max_seq, scale); \
    if (d == 64 && nw >= 16) return small ? launch_decode_t<CT, 64, 4, 16>(L, q, k_cache, v_cache, st, out, sc, H, Hkv, max_seq, scale)   \
                              : launch_decode_t<CT, 64, 8, 8>(L, q, k_cache, v_cache, st, out, sc, H, Hkv, max_seq, scale); \
    if (d == 128) return small ? launch_decode_t<CT, 128, 4, 4>(L, q, k_cache, v_cache, st, out, sc, H, Hkv, max_seq, scale) \
                               : launch_decode_t<CT, 128, 8, 4>(L, q, k_cache, v_cache, st, out, sc, H, Hkv, max_seq, scale); \
    if (d == 64) return small ? launch_decode_t<CT, 64, 4, 4>(L, q, k_cache, v_cache, st, out, sc, H, Hkv, max_seq, scale)   \
                              : launch_decode_t<CT, 64, 8, 4>(L, q, k_cache, v_cache, st, out, sc, H, Hkv, max_seq, scale);
    if (dtype == FL_DTYPE_BF16) { FL_DISPATCH(bf16_t) }
    else { FL_DISPATCH(float) }
#undef FL_DISPATCH
    FL_FAIL(FL_ERR_UNSUPPORTED, "attention: head_dim %lld not supported (64 or 128)", (long long)d);
}

// ------------------------------------------------------------------------------- prefill (T > 1)
// grid (Hkv, T, ceil(G/GMAX)): one workgroup per (kv head, query row).  Mask (App. A.5): the
// cached prefix [0,len) is fully visible; in-call key j is visible to query t iff j <= t and
// (no window or j + window >= t).
template <typename CT, int D, int GMAX>
__global__ __launch_bounds__(256) void attn_prefill_kernel(const CT *__restrict__ q, const CT *__restrict__ kc,
                                                           const CT *__restrict__ vc, const StepState *__restrict__ st,
                                                           CT *__restrict__ out, int T, int H, int Hkv, int max_seq,
                                                           float scale, int window) {
    __shared__ float lds[4 * GMAX * (D + 2)];
    const int hk = blockIdx.x, t = blockIdx.y;
    const int Gall = H / Hkv, g0 = blockIdx.z * GMAX;
    const int G = min(GMAX, Gall - g0);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int len = (int)st->len;
    const int hq0 = hk * Gall + g0;
    const CT *kb = kc + (size_t)hk * max_seq * D, *vb = vc + (size_t)hk * max_seq * D;

    float qv[GMAX][8];
    load_q<CT, D, GMAX>(qv, q + (size_t)t * H * D, hq0, G, lane, scale);
    AttnState<GMAX> s; s.init();
    // prefix cached before the API call: unmasked; window runs over the call's own tokens (call0 <= len)
    const int c0 = (int)st->call0;
    int lo = c0;
    if (window >= 0 && len + t - window > c0) lo = len + t - window;
    if (lo == c0) {
        attend_range<CT, D, GMAX, 4, 2>(s, qv, G, kb, vb, 0, len + t + 1, wave, lane);
    } else {
        if (c0 > 0) attend_range<CT, D, GMAX, 4, 2>(s, qv, G, kb, vb, 0, c0, wave, lane);
        attend_range<CT, D, GMAX, 4, 2>(s, qv, G, kb, vb, lo, len + t + 1, wave, lane);
    }
    merge_to_lds<D, GMAX, 4>(s, G, lds, wave, lane);
    for (int e = threadIdx.x; e < G * (D / 4); e += 256) {
        const int g = e / (D / 4), j4 = (e % (D / 4)) * 4;
        float M, L, O[4];
        combine_lds<D, GMAX, 4>(lds, g, j4, M, L, O);
        const float inv = 1.0f / L;
#pragma unroll
        for (int j = 0; j < 4; j++) elem<CT>::st(out + ((size_t)t * H + hq0 + g) * D + j4 + j, O[j] * inv);
    }
}

template <typename CT, int D, int GMAX>
static int launch_prefill_t(Launcher &L, const void *q, const void *kc, const void *vc, const StepState *st, void *out,
                            int64_t T, int64_t H, int64_t Hkv, int64_t max_seq, float scale, int64_t window) {
    const int G = (int)(H / Hkv);
    dim3 grid((unsigned)Hkv, (unsigned)T, (unsigned)((G + GMAX - 1) / GMAX));
    double flops = 2.0 * (double)T * T * H * D;       // QK^T + PV over the causal half
    return L.launch(KC_ATTN_PREFILL, 0, flops, attn_prefill_kernel<CT, D, GMAX>, grid, dim3(256), 0, (const CT *)q,
                    (const CT *)kc, (const CT *)vc, st, (CT *)out, (int)T, (int)H, (int)Hkv, (int)max_seq, scale,
                    (int)window);
}

int launch_attn_prefill(Launcher &L, int dtype, const void *q, const void *k_cache, const void *v_cache,
                        const StepState *st, void *out, int64_t T, int64_t H, int64_t Hkv, int64_t d,
                        int64_t max_seq, float scale, int64_t window) {
    const int G = (int)(H / Hkv);
    const bool small = G <= 4;
#define FL_DISPATCH(CT)                                                                                             \
    if (d == 128) return small ? launch_prefill_t<CT, 128, 4>(L, q, k_cache, v_cache, st, out, T, H, Hkv, max_seq, scale, window) \
                               : launch_prefill_t<CT, 128, 8>(L, q, k_cache, v_cache, st, out, T, H, Hkv, max_seq, scale, window); \
    if (d == 64) return small ? launch_prefill_t<CT, 64, 4>(L, q, k_cache, v_cache, st, out, T, H, Hkv, max_seq, scale, window)   \
                              : launch_prefill_t<CT, 64, 8>(L, q, k_cache, v_cache, st, out, T, H, Hkv, max_seq, scale, window);
    if (dtype == FL_DTYPE_BF16) { FL_DISPATCH(bf16_t) }
    else { FL_DISPATCH(float) }
#undef FL_DISPATCH
    FL_FAIL(FL_ERR_UNSUPPORTED, "attention: head_dim %lld not supported (64 or 128)", (long long)d);
}

}  // namespace fl
