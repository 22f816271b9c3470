"""Token selection (row N3 of the scope table): LogitsProcessor::new(0, Some(temperature), None).sample().

CPU part.  The oracle (oracle/ref_sampler.c) restates candle's LogitsProcessor over rand 0.8's StdRng /
WeightedIndex; its ChaCha core is pinned by the published zero-key ChaCha20 keystream (RFC 7539 section
2.3 construction, djb's original 64-bit-counter layout), the rest is "parity unpinned" (no vector in the
reference).  The host mirror's LogitsProcessor (fastllm_amd/host/fastllm_host.hpp) is the same sequential
algorithm written independently in C++: the two must agree bit for bit.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle
from test_host_mirror import host  # noqa: F401

# first block of the ChaCha20 keystream for the all-zero key and nonce ("76 b8 e0 ad a0 f1 3d 90 ...")
CHACHA20_ZERO = bytes.fromhex("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                              "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")


def test_chacha_core_known_answer():
    words = oracle.chacha_block([0] * 8, 0, 20)
    assert b"".join(int(w).to_bytes(4, "little") for w in words) == CHACHA20_ZERO
    # 12 rounds and the block counter change the output (the stream StdRng actually uses)
    assert oracle.chacha_block([0] * 8, 0, 12) != words
    assert oracle.chacha_block([0] * 8, 1, 12) != oracle.chacha_block([0] * 8, 0, 12)


def test_seed_expansion_is_pcg32():
    """seed_from_u64: eight PCG32 XSH-RR outputs of the LCG started at the seed."""
    mask = (1 << 64) - 1
    for seed in (0, 1, 299792458, mask):
        st, key = seed, []
        for _ in range(8):
            st = (st * 6364136223846793005 + 11634580027462260723) & mask
            xs = (((st >> 18) ^ st) >> 27) & 0xFFFFFFFF
            rot = st >> 59
            key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & 0xFFFFFFFF)
        assert oracle.Sampler(seed, 1.0).key == key


def test_rng_stream_is_block_ordered():
    s = oracle.Sampler(7, 1.0)
    words = [s.next_u32() for _ in range(40)]
    blocks = oracle.chacha_block(s.key, 0, 12) + oracle.chacha_block(s.key, 1, 12) + oracle.chacha_block(s.key, 2, 12)
    assert words == blocks[:40] and s.draws == 40


def test_temperature_threshold_and_argmax_ties():
    lg = np.array([0.5, 2.0, -1.0, 2.0, 1.0], dtype=np.float32)
    for t in (None, 0.0, 9e-8):
        s = oracle.Sampler(0, t)
        assert s.sample(lg) == 3 and s.draws == 0            # ArgMax: last maximal index, no random word used
    s = oracle.Sampler(0, 1e-7)
    s.sample(lg)
    assert s.draws == 1                                       # at the threshold it is Sampling::All


def test_sampling_distribution_and_boundaries():
    rs = np.random.RandomState(3)
    lg = (rs.randn(50) * 2).astype(np.float32)
    s = oracle.Sampler(0, 0.7)
    n = 20000
    counts = np.zeros(50)
    for _ in range(n):
        tok, info = s.sample(lg, want_info=True)
        assert info["cum_lo"] <= info["chosen"] < info["cum_hi"] or tok == 49
        assert 0.0 <= info["chosen"] < info["total"]
        counts[tok] += 1
    p = np.exp(lg.astype(np.float64) / 0.7)
    p /= p.sum()
    # chi-square against the softmax (49 dof: 99.9th percentile ~ 85)
    chi2 = ((counts - n * p) ** 2 / (n * p)).sum()
    assert chi2 < 110, chi2


def _lp(host, seed, temperature):
    h = C.c_void_p()
    host.flh_logits_processor_new.argtypes = [C.c_uint64, C.c_int, C.c_double, C.POINTER(C.c_void_p)]
    host.flh_logits_processor_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
    host.flh_logits_processor_free.argtypes = [C.c_void_p]
    host.flh_logits_processor_free.restype = None
    assert host.flh_logits_processor_new(seed, 0 if temperature is None else 1, temperature or 0.0, C.byref(h)) == 0
    return h


@pytest.mark.parametrize("seed,temperature,V", [(0, 0.8, 320), (0, 1.0, 32000), (5, 0.3, 1000), (0, None, 777), (9, 2.5, 152064)])
def test_host_mirror_logits_processor_matches_oracle(host, seed, temperature, V):
    if not os.path.exists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fastllm_amd", "lib", "libfastllm_host.so")):
        pytest.skip("host library not built")
    h = _lp(host, seed, temperature)
    o = oracle.Sampler(seed, temperature)
    rs = np.random.RandomState(V)
    try:
        for step in range(25):
            lg = (rs.randn(V) * (1.0 + step % 4)).astype(np.float32)
            tok, draws = C.c_uint32(0), C.c_uint64(0)
            assert host.flh_logits_processor_sample(h, lg.ctypes.data, V, C.byref(tok), C.byref(draws)) == 0
            assert tok.value == o.sample(lg) and draws.value == o.draws
    finally:
        host.flh_logits_processor_free(h)
