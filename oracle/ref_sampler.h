/*
 * oracle/ref_sampler.h -- CPU restatement of LogitsProcessor::sample as the reference uses it
 * (mod.rs:157-158,308-310,373-374,425-428).  TEST INFRASTRUCTURE ONLY; "parity unpinned" -- see ref_sampler.c.
 */
#ifndef ORACLE_REF_SAMPLER_H
#define ORACLE_REF_SAMPLER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_rng { uint32_t key[8]; uint64_t word; } orc_rng;      /* StdRng: ChaCha12 key + words consumed */
typedef struct orc_sampler { orc_rng rng; double temperature; int argmax; } orc_sampler;
/* what the draw looked like, for margin-aware comparison with a parallel implementation */
typedef struct orc_sample_info { uint32_t u; float chosen, total, cum_lo, cum_hi, p; } orc_sample_info;

void     orc_chacha_block(const uint32_t key[8], uint64_t counter, int rounds, uint32_t out[16]);
void     orc_rng_seed_from_u64(orc_rng *g, uint64_t seed);
uint32_t orc_rng_next_u32(orc_rng *g);
/* temperature < 0 stands for None */
void     orc_sampler_init(orc_sampler *s, uint64_t seed, double temperature);
/* scratch: n floats */
uint32_t orc_sample(orc_sampler *s, const float *logits, size_t n, float *scratch, orc_sample_info *info);

#ifdef __cplusplus
}
#endif
#endif
