/*
 * oracle/ref_sampler.c -- CPU restatement of the token-selection step of the reference's generate loop.
 *
 * TEST INFRASTRUCTURE ONLY (see ref_forward.h): only tests/ may call this, as the checker.
 * PARITY STATUS: "parity unpinned".  The reference does
 *     LogitsProcessor::new(Default::default(), Some(temperature as f64), None)      [seed 0]
 *         /root/reference/src/models/mod.rs:157-158,183-184,209-210,373-374
 *     logits_processor.sample(&last_logits)            mod.rs:308-310,425-428
 * and all of the arithmetic lives in un-vendored third-party crates that are absent from /root/reference
 * (Cargo.lock is git-ignored, so the patch versions are unknown):
 *   candle-transformers ^0.8.2  generation::LogitsProcessor
 *       temperature < 1e-7 (or None)  -> Sampling::ArgMax: iter().enumerate().max_by(total_cmp) (last max wins)
 *       otherwise, top_p None          -> Sampling::All: prs = softmax_last_dim(logits.to_dtype(F32) / temperature),
 *                                         then sample_multinomial(prs) = WeightedIndex::new(prs).sample(&mut rng)
 *       `Tensor / f64` is affine(1/temperature, 0): the f64 reciprocal is cast to f32 and multiplied
 *   candle-nn ^0.8.2  ops::softmax_last_dim (CPU): max, exp(x - max), sum, divide -- all f32
 *   rand 0.8.5  rngs::StdRng = rand_chacha::ChaCha12Rng;  SeedableRng::seed_from_u64 expands the u64 with PCG32
 *       (MUL 6364136223846793005, INC 11634580027462260723, XSH-RR output) into the 32-byte key, little endian;
 *       the stream is ChaCha with 12 rounds, 64-bit block counter from 0, stream id 0, consumed one u32 word
 *       at a time in block order;
 *       distributions::WeightedIndex<f32>: cumulative weights summed left to right in f32 (n-1 of them kept),
 *       chosen = Uniform::new(0, total).sample(rng), index = partition_point(|w| w <= chosen);
 *       UniformFloat<f32>::sample: value1_2 = f32::from_bits((next_u32() >> 9) | 0x3f80_0000);
 *       (value1_2 - 1.0) * scale + low, with scale = total (decreased by ulps while scale*max_rand+low >= high).
 * All of the above is restated from the published algorithms ([UPSTREAM-RECALLED]); the
 * ChaCha core is pinned by the RFC 7539 / djb zero-key ChaCha20 keystream (tests/test_sampler.py), the rest has
 * no vector in the reference's tests to pin it.
 */
#include "ref_sampler.h"

#include <math.h>
#include <string.h>

static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define QR(a, b, c, d) \
    do { a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); \
         a += b; d ^= a; d = rotl32(d, 8);  c += d; b ^= c; b = rotl32(b, 7); } while (0)

void orc_chacha_block(const uint32_t key[8], uint64_t counter, int rounds, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};      /* "expand 32-byte k" */
    for (int i = 0; i < 8; i++) s[4 + i] = key[i];
    s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = 0; s[15] = 0;
    uint32_t x[16];
    memcpy(x, s, sizeof x);
    for (int r = 0; r < rounds; r += 2) {
        QR(x[0], x[4], x[8], x[12]);  QR(x[1], x[5], x[9], x[13]);
        QR(x[2], x[6], x[10], x[14]); QR(x[3], x[7], x[11], x[15]);
        QR(x[0], x[5], x[10], x[15]); QR(x[1], x[6], x[11], x[12]);
        QR(x[2], x[7], x[8], x[13]);  QR(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
}

void orc_rng_seed_from_u64(orc_rng *g, uint64_t state) {
    for (int i = 0; i < 8; i++) {
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        g->key[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));       /* rotate_right; LE bytes = LE word */
    }
    g->word = 0;
}

uint32_t orc_rng_next_u32(orc_rng *g) {
    uint32_t blk[16];
    orc_chacha_block(g->key, g->word / 16, 12, blk);
    return blk[g->word++ % 16];
}

void orc_sampler_init(orc_sampler *s, uint64_t seed, double temperature) {
    orc_rng_seed_from_u64(&s->rng, seed);
    s->argmax = !(temperature >= 1e-7);          /* None, or Some(v) with v < 1e-7 */
    s->temperature = temperature;
}

uint32_t orc_sample(orc_sampler *s, const float *logits, size_t n, float *scratch, orc_sample_info *info) {
    if (info) memset(info, 0, sizeof *info);
    if (s->argmax) {
        size_t best = 0;
        for (size_t i = 1; i < n; i++) if (!(logits[i] < logits[best])) best = i;      /* last max wins */
        return (uint32_t)best;
    }
    const float mul = (float)(1.0 / s->temperature);
    float mx = -INFINITY;
    for (size_t i = 0; i < n; i++) { scratch[i] = logits[i] * mul; if (scratch[i] > mx) mx = scratch[i]; }
    float sum = 0.f;
    for (size_t i = 0; i < n; i++) { scratch[i] = expf(scratch[i] - mx); sum += scratch[i]; }
    for (size_t i = 0; i < n; i++) scratch[i] /= sum;
    /* WeightedIndex::new */
    float total = scratch[0];
    for (size_t i = 1; i < n; i++) total += scratch[i];
    /* UniformFloat::new(0, total) */
    const float max_rand = 1.0f - 1.1920929e-07f;                                     /* 1 - 2^-23 */
    float scale = total;
    while (scale * max_rand + 0.0f >= total) { uint32_t b; memcpy(&b, &scale, 4); b -= 1; memcpy(&scale, &b, 4); }
    /* sample */
    const uint32_t u = orc_rng_next_u32(&s->rng);
    const uint32_t bits = (u >> 9) | 0x3f800000u;
    float v12; memcpy(&v12, &bits, 4);
    const float chosen = (v12 - 1.0f) * scale + 0.0f;
    /* partition_point over the n-1 cumulative weights [w0, w0+w1, ...] */
    float cum = scratch[0], lo = 0.f;
    size_t idx = 0;
    while (idx < n - 1 && cum <= chosen) { lo = cum; idx++; cum += scratch[idx]; }
    if (info) { info->u = u; info->chosen = chosen; info->total = total; info->cum_lo = lo; info->cum_hi = cum; info->p = scratch[idx]; }
    return (uint32_t)idx;
}
