// gemm_w4.h -- pieces shared by the four-wave prefill GEMM kernels (k_gemm_8p.hip: 256 x 256 tile; k_gemm_h4.hip: 128 x 256
// tile with the K slices summed inside the launch): LDS-DMA pieces as buffer loads, fragment reads, and the accumulator
// registers a[0:255] owned by inline asm.
#pragma once
#include <type_traits>
#include <utility>

#include "kernels.h"

namespace fl {

constexpr int P_BM = 256, P_BN = 256, P_BK = 64;
constexpr int P_HALF = 128 * P_BK * 2;                 // 16 KiB
// tile row of local row r of half h:  A halves: M group r>>6, 64 rows each (the B halves of the four-wave kernels likewise)
__device__ inline int a_row(int h, int r) { return (r >> 6) * 128 + h * 64 + (r & 63); }

typedef int int4w __attribute__((ext_vector_type(4)));
constexpr int W4_BIAS = 3072;
__device__ inline void w4_set_m0(unsigned m0v) { asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" : : "s"(m0v) : "memory"); }
// NT: the non-temporal form, for bytes no other workgroup will ask for again (tools/micro/ingest_bench.hip: an HBM stream through
// LDS-DMA runs at 5.97 TB/s plain and 6.89 with nt; bytes another workgroup of the XCD re-reads should stay plain: they are its L2 hits)
template <int S, bool NT = false>
__device__ inline void w4_piece(int4w rsrc, unsigned voff, unsigned soff) {
    if constexpr (NT) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%c3 nt lds" : : "v"(voff), "s"(rsrc), "s"(soff), "i"(S * 1024) : "memory");
    else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%c3 lds" : : "v"(voff), "s"(rsrc), "s"(soff), "i"(S * 1024) : "memory");
}
__device__ inline int4w w4_rsrc(const void *base) {                     // raw buffer over [base - 3072, +4 GiB): no stride, no swizzle
    const unsigned long long b = (unsigned long long)base - W4_BIAS;
    return int4w{(int)(unsigned)b, (int)(unsigned)(b >> 32) & 0xFFFF, -1, 0x00020000};
}

// The 256 accumulator registers are a[0:255], owned by the asm statements below (hipcc's allocator, given 64 accumulator tiles
// next to 128 fragment registers, shuffles them through v_accvgpr moves and scratch: 568 moves and 150 scratch accesses per
// two K tiles).  Tile (n half h, row block i, column block j) lives in a[16 (8 h + i) + 4 j ..+3].  The compiler sees none of
// them: build check = no v_accvgpr_* outside these statements and no scratch in the kernel (tests/test_dot_hazard.py::test_four_wave_gemm_owns_its_accumulator_registers).
#define W4_AGPRS "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","a128","a129","a130","a131","a132","a133","a134","a135","a136","a137","a138","a139","a140","a141","a142","a143","a144","a145","a146","a147","a148","a149","a150","a151","a152","a153","a154","a155","a156","a157","a158","a159","a160","a161","a162","a163","a164","a165","a166","a167","a168","a169","a170","a171","a172","a173","a174","a175","a176","a177","a178","a179","a180","a181","a182","a183","a184","a185","a186","a187","a188","a189","a190","a191","a192","a193","a194","a195","a196","a197","a198","a199","a200","a201","a202","a203","a204","a205","a206","a207","a208","a209","a210","a211","a212","a213","a214","a215","a216","a217","a218","a219","a220","a221","a222","a223","a224","a225","a226","a227","a228","a229","a230","a231","a232","a233","a234","a235","a236","a237","a238","a239","a240","a241","a242","a243","a244","a245","a246","a247","a248","a249","a250","a251","a252","a253","a254","a255"
__device__ inline int4w frag4(const unsigned char *half, int row, int chunk) {
    const int pc = chunk ^ ((row >> 1) & 7);
    return *reinterpret_cast<const int4w *>(half + row * 128 + pc * 16);
}
template <int A0>                                                       // acc tile a[A0:A0+3] += X-fragment . W-fragment over K = 32
__device__ inline void w4_mfma(int4w x, int4w w) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" : : "v"(x), "v"(w), "i"(A0), "i"(A0 + 3));
}
template <int A0> __device__ inline void w4_zero16() {                  // a[A0:A0+15] = 0
    asm volatile("v_accvgpr_write_b32 a[%c0], 0\n\tv_accvgpr_write_b32 a[%c1], 0\n\tv_accvgpr_write_b32 a[%c2], 0\n\tv_accvgpr_write_b32 a[%c3], 0\n\t"
                 "v_accvgpr_write_b32 a[%c4], 0\n\tv_accvgpr_write_b32 a[%c5], 0\n\tv_accvgpr_write_b32 a[%c6], 0\n\tv_accvgpr_write_b32 a[%c7], 0\n\t"
                 "v_accvgpr_write_b32 a[%c8], 0\n\tv_accvgpr_write_b32 a[%c9], 0\n\tv_accvgpr_write_b32 a[%c10], 0\n\tv_accvgpr_write_b32 a[%c11], 0\n\t"
                 "v_accvgpr_write_b32 a[%c12], 0\n\tv_accvgpr_write_b32 a[%c13], 0\n\tv_accvgpr_write_b32 a[%c14], 0\n\tv_accvgpr_write_b32 a[%c15], 0"
                 : : "i"(A0), "i"(A0 + 1), "i"(A0 + 2), "i"(A0 + 3), "i"(A0 + 4), "i"(A0 + 5), "i"(A0 + 6), "i"(A0 + 7), "i"(A0 + 8),
                     "i"(A0 + 9), "i"(A0 + 10), "i"(A0 + 11), "i"(A0 + 12), "i"(A0 + 13), "i"(A0 + 14), "i"(A0 + 15));
}
template <int A0> __device__ inline float4v w4_read() {                 // (behind the s_nop that ends the K loop)
    float a, b, c, d;
    asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
                 : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "i"(A0), "i"(A0 + 1), "i"(A0 + 2), "i"(A0 + 3));
    return float4v{a, b, c, d};
}
template <int... Is, class F> __device__ inline void w4_for_impl(std::integer_sequence<int, Is...>, F &&f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ inline void w4_for(F &&f) { w4_for_impl(std::make_integer_sequence<int, N>{}, f); }

// row scale of output row `row`: 1/rms from a residual epilogue's partial sums (kernels.h, RsParts), the caller's vector, or 1
__device__ inline float row_scale_of(const float *__restrict__ row_scale, const RsParts &rsp, int row) {
    if (rsp.part) {
        const float4v *p = reinterpret_cast<const float4v *>(rsp.part + (size_t)row * rsp.np);
        float ss = 0.f;
        for (int i = 0; i < rsp.np / 4; i++) { const float4v v = p[i]; ss += v[0]; ss += v[1]; ss += v[2]; ss += v[3]; }
        return 1.0f / sqrtf(ss * rsp.inv_h + rsp.eps);
    }
    return row_scale ? row_scale[row] : 1.0f;
}

// ---- epilogue of one wave's 128 x 64 block, shared by the GEMM kernel and the stream-K fix-up kernel.
// C/D map of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.  The row scales of the tile go through LDS (one global
// load per row, not one per accumulator row per lane); whole tiles take a path without bounds checks (the checked one spends a
// branch pair per store: ~6 us of VALU per tile, tools/stamps_8p.py).
struct EpiCtx {
    void *out; const float *rs_rows;          // output base (slab applied); LDS row scales of this wave's rows (+ 16 i + reg)
    int T, N, epi, ldc, mw, nw, cn, tn, wc;   // mw / nw: first row / this lane's first column of the wave's block
    float bj[4], wj[4];                       // bias / next-norm weight of the lane's four columns
    ResidEpi re;
};
__device__ inline void epi_ctx_init(EpiCtx &c, void *out, const float *bias, const float *rs_lds, int T, int N, int epi, int ldc,
                                    int m0, int n0, int tn, int wr, int wc, int tid_e, const ResidEpi &re, int wave_rows = 128) {
    const int cn = tid_e & 15, rm = ((tid_e >> 4) & 3) * 4;
    c.out = out; c.rs_rows = rs_lds + wr * wave_rows + rm; c.T = T; c.N = N; c.epi = epi; c.ldc = ldc;
    c.mw = m0 + wr * wave_rows + rm; c.nw = n0 + wc * 64 + cn; c.cn = cn; c.tn = tn; c.wc = wc; c.re = re;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        c.bj[j] = bias && epi != EPI_GATEUP && c.nw + j * 16 < N ? bias[c.nw + j * 16] : 0.f;
        c.wj[j] = epi == EPI_RESID && c.nw + j * 16 < N ? re.w[c.nw + j * 16] : 0.f;
    }
}
// rows [16 i, 16 i + 16) of the wave's block: v[j] = the accumulator tile of column block j
// CHK: 0 whole tile, no bounds checks; 1 rows and columns checked; 2 rows only (a ragged last row tile of a matrix whose width is whole
// tiles: the usual ragged case -- one compare per accumulator row instead of a branch pair per store)
// JMAX: column blocks of the group that exist (the fp32 and gate/up epilogues; a 224-column tile's last group is two blocks wide)
template <int CHK, int JMAX = 4>
__device__ inline void store_rows(const EpiCtx &c, int i, const float4v (&v)[4]) {
    const int T = c.T, N = c.N, ldc = c.ldc, mw = c.mw, nw = c.nw;
    const float4v rs4 = *reinterpret_cast<const float4v *>(c.rs_rows + i * 16);
    if (c.epi == EPI_GATEUP) {
        bf16_t *ob = reinterpret_cast<bf16_t *>(c.out) + (size_t)(mw + i * 16) * (ldc / 2) + (nw >> 5) * 16 + c.cn;   // j = 2: + 16
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            if (CHK && mw + i * 16 + rg >= T) continue;
#pragma unroll
            for (int j = 0; j < JMAX; j += 2) {
                if (CHK == 1 && nw + j * 16 + 16 >= N) continue;                   // gate column nw + 16 j, up 16 further
                const float gt = v[j][rg] * rs4[rg], up = v[j + 1][rg] * rs4[rg];
                // silu(g) * u: v_exp + v_rcp (1 ulp each); the result is rounded to bf16
                const float av = gt * up * __builtin_amdgcn_rcpf(1.0f + __expf(-gt));
                ob[(size_t)rg * (ldc / 2) + j * 8] = float_to_bf16_bits(av);
            }
        }
    } else if (c.epi == EPI_RESID) {
        // residual add + the next norm's x * w + this wave's share of the row sums of squares (lanes of a row: cn).
        // All sixteen h values of the row block are requested before the first is used (the partial-sum stores are
        // floats too: without the explicit order every row would wait for its own round trip).
        float *__restrict__ hb = c.re.h + (size_t)(mw + i * 16) * ldc + nw;
        bf16_t *__restrict__ xb = reinterpret_cast<bf16_t *>(c.re.xn) + (size_t)(mw + i * 16) * ldc + nw;
        float hv[4][4], ss[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool ok = !CHK || (mw + i * 16 + rg < T && (CHK == 2 || nw + j * 16 < N));
                hv[rg][j] = ok ? hb[(size_t)rg * ldc + j * 16] : 0.f;
            }
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            ss[rg] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (CHK && (mw + i * 16 + rg >= T || (CHK == 1 && nw + j * 16 >= N))) continue;
                const float hn = hv[rg][j] + (v[j][rg] * rs4[rg] + c.bj[j]);
                hb[(size_t)rg * ldc + j * 16] = hn;
                xb[(size_t)rg * ldc + j * 16] = float_to_bf16_bits(hn * c.wj[j]);
                ss[rg] = fmaf(hn, hn, ss[rg]);
            }
        }
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            float sr = ss[rg];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sr += __shfl_xor(sr, o);
            if ((!CHK || mw + i * 16 + rg < T) && c.cn == 0) c.re.part[(size_t)(mw + i * 16 + rg) * c.re.np + c.tn * 4 + c.wc] = sr;
        }
    } else {
        float *ob = reinterpret_cast<float *>(c.out) + (size_t)(mw + i * 16) * ldc + nw;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
            if (CHK && mw + i * 16 + rg >= T) continue;
#pragma unroll
            for (int j = 0; j < JMAX; j++) {
                if (CHK == 1 && nw + j * 16 >= N) continue;
                ob[(size_t)rg * ldc + j * 16] = v[j][rg] * rs4[rg] + c.bj[j];
            }
        }
    }
}

// ---- EPI_QKV_ROPE: the QKV projection's epilogue does what rope_kv_vec_kernel does behind a plain fp32 output -- row scale, bias,
// rotate-half RoPE of the q and k heads, q to the activation buffer, k / v appended to the cache (v transposed for the MFMA
// attention) -- so the fp32 QKV matrix never exists and the launch goes away (candle_nn::rotary_emb::rope, App. A.4; K5 in place).
// A head's columns belong to ONE wave: d = 128: the wave's 128 columns = n half 0 | n half 1 (blocks b and b + 4: one owner for 1,
// 2 or 4 slices); d = 64: one n half.  The accumulator layout (a lane: 4 rows x one column per 16-column block) would make 2-byte
// stores and one cos / sin load per element (measured: 16 us per launch at 512 tokens), so the wave's 16 x d part goes through a
// wave-private LDS tile (fp32, padded rows: conflict-free both ways) and comes back ROW-major: a lane then holds 8 consecutive
// columns of a row and their 8 partners, reads cos / sin as float4s and stores 16 bytes per half -- or, for the transposed value
// cache, 8 consecutive tokens of one column.
struct RopeLane {
    const RopeEpi *ro; const float *rs_w; const float *bias;      // rs_w: LDS row scales of this wave's rows
    float *wlds;                                                   // this wave's staging tile: 16 x (128 + 4) floats
    int T, N, t0w, lane; uint32_t pos0, len;
};
__device__ inline uint4v pack8(const float (&y)[8]) {
    return uint4v{(uint32_t)float_to_bf16_bits(y[0]) | ((uint32_t)float_to_bf16_bits(y[1]) << 16), (uint32_t)float_to_bf16_bits(y[2]) | ((uint32_t)float_to_bf16_bits(y[3]) << 16),
                  (uint32_t)float_to_bf16_bits(y[4]) | ((uint32_t)float_to_bf16_bits(y[5]) << 16), (uint32_t)float_to_bf16_bits(y[6]) | ((uint32_t)float_to_bf16_bits(y[7]) << 16)};
}
// rows [16 i, 16 i + 16) of a head of width D (128: lo = n half 0, hi = n half 1; 64: lo only, hi unused); col0 = the head's first column
template <int D>
__device__ inline void rope_head(const RopeLane &c, int i, int col0, const float4v (&lo)[4], const float4v (&hi)[4]) {
    constexpr int HALF = D / 2, LD = D + 4, CH = HALF / 8;        // CH: 8-column chunks per half row
    const RopeEpi &ro = *c.ro;
    if (col0 >= c.N) return;                                       // (wave-uniform: N is a multiple of the head width)
    float *w = c.wlds;
    {
        const int cn = c.lane & 15, rg0 = ((c.lane >> 4) & 3) * 4;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int rg = 0; rg < 4; rg++) {
                w[(rg0 + rg) * LD + j * 16 + cn] = lo[j][rg];
                if (D == 128) w[(rg0 + rg) * LD + 64 + j * 16 + cn] = hi[j][rg];
            }
    }
    asm volatile("" ::: "memory");                                 // (one wave, in-order LDS: the tile is complete for the reads below)
    const int hd = (col0 + ro.col_base) / D, tb = c.t0w + i * 16;
    const bool rot = hd < ro.H + ro.Hkv;
    if (rot || !ro.v_transposed) {
        // q / k (rotated) or a row-major value cache: 16 rows x CH chunks of 8 (column, partner) pairs
#pragma unroll
        for (int u = 0; u < (16 * CH + 63) / 64; u++) {
            const int item = c.lane + 64 * u, row = item / CH, c8 = (item % CH) * 8, t = tb + row;
            if (item >= 16 * CH) break;
            const float rs = c.rs_w[i * 16 + row];
            float x0[8], x1[8], cs[8], sn[8], y0[8], y1[8];
            *reinterpret_cast<float4v *>(x0) = *reinterpret_cast<const float4v *>(w + row * LD + c8);
            *reinterpret_cast<float4v *>(x0 + 4) = *reinterpret_cast<const float4v *>(w + row * LD + c8 + 4);
            *reinterpret_cast<float4v *>(x1) = *reinterpret_cast<const float4v *>(w + row * LD + HALF + c8);
            *reinterpret_cast<float4v *>(x1 + 4) = *reinterpret_cast<const float4v *>(w + row * LD + HALF + c8 + 4);
            if (rot) {
                const uint32_t pos = c.pos0 + (uint32_t)t, p = pos < (uint32_t)ro.max_pos ? pos : (uint32_t)ro.max_pos - 1;   // host validates range
                const float *ct = ro.cos_tab + (size_t)p * HALF + c8, *st = ro.sin_tab + (size_t)p * HALF + c8;
                *reinterpret_cast<float4v *>(cs) = *reinterpret_cast<const float4v *>(ct); *reinterpret_cast<float4v *>(cs + 4) = *reinterpret_cast<const float4v *>(ct + 4);
                *reinterpret_cast<float4v *>(sn) = *reinterpret_cast<const float4v *>(st); *reinterpret_cast<float4v *>(sn + 4) = *reinterpret_cast<const float4v *>(st + 4);
            }
#pragma unroll
            for (int e = 0; e < 8; e++) {
                const float a = x0[e] * rs + (c.bias ? c.bias[col0 + c8 + e] : 0.f), b = x1[e] * rs + (c.bias ? c.bias[col0 + HALF + c8 + e] : 0.f);
                if (rot) rope_rotate(a, b, cs[e], sn[e], y0[e], y1[e]);
                else { y0[e] = a; y1[e] = b; }
            }
            if (t < c.T) {
                bf16_t *o = hd < ro.H ? reinterpret_cast<bf16_t *>(ro.q_out) + ((size_t)t * ro.H + hd) * D
                          : hd < ro.H + ro.Hkv ? reinterpret_cast<bf16_t *>(ro.k_cache) + ((size_t)(hd - ro.H) * ro.max_seq + c.len + t) * D
                                               : reinterpret_cast<bf16_t *>(ro.v_cache) + ((size_t)(hd - ro.H - ro.Hkv) * ro.max_seq + c.len + t) * D;
                *reinterpret_cast<uint4v *>(o + c8) = pack8(y0);
                *reinterpret_cast<uint4v *>(o + HALF + c8) = pack8(y1);
            }
        }
    } else {
        // transposed value cache [Hkv][D][max_seq]: a lane takes one column and 8 consecutive tokens
        const int hv = hd - ro.H - ro.Hkv;
#pragma unroll
        for (int u = 0; u < (2 * D) / 64; u++) {
            const int item = c.lane + 64 * u, col = item >> 1, r8 = (item & 1) * 8, t = tb + r8;
            const float bcol = c.bias ? c.bias[col0 + col] : 0.f;
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; e++) y[e] = w[(r8 + e) * LD + col] * c.rs_w[i * 16 + r8 + e] + bcol;
            bf16_t *o = reinterpret_cast<bf16_t *>(ro.v_cache) + ((size_t)hv * D + col) * ro.max_seq + c.len + t;
            if (t + 7 < c.T && ((c.len + t) & 7) == 0) {
                *reinterpret_cast<uint4v *>(o) = pack8(y);
            } else {
#pragma unroll
                for (int e = 0; e < 8; e++)
                    if (t + e < c.T) o[e] = float_to_bf16_bits(y[e]);
            }
        }
    }
    asm volatile("" ::: "memory");                                 // (the next call rewrites the tile behind these reads: in-order LDS)
}
__device__ inline void rope_rows128(const RopeLane &c, int i, int colw, const float4v (&lo)[4], const float4v (&hi)[4]) { rope_head<128>(c, i, colw, lo, hi); }
__device__ inline void rope_rows64(const RopeLane &c, int i, int colh, const float4v (&v)[4]) { rope_head<64>(c, i, colh, v, v); }

}  // namespace fl
