"""Pin the CPU restatement (oracle/) against the committed golden vectors.

The vectors come from HuggingFace transformers fp32/eager (tests/golden/make_golden.py);
the reference itself has no forward-pass fixtures -> "parity unpinned"
at the candle boundary, pinned against an independent implementation instead.
"""
import json
import os

import numpy as np
import pytest

import synth
from oracle import oracle

CASES = ["llama_a", "llama_mha", "mistral_a", "mistral_win", "qwen2_a", "qwen2_win", "llama_d100", "qwen2_d96", "mistral_d48"]
TOL = 2e-4     # fp32 vs fp32, different summation order


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


@pytest.fixture(scope="module", params=CASES)
def case(request, golden_dir):
    z, meta = load(golden_dir, request.param)
    cfg = synth.CONFIGS[request.param]
    assert meta["config"] == json.loads(json.dumps(cfg))
    w = synth.synth_weights(cfg)
    assert synth.weights_digest(w) == meta["weights_sha256"], "synthetic weight stream drifted"
    m = oracle.OracleModel(cfg, synth.as_f32(w))
    return z, meta, cfg, m


def test_prefill_logits(case):
    z, meta, cfg, m = case
    c = m.new_cache(64)
    lg = m.forward(c, z["prompt"], 0)
    assert len(c) == meta["T"]
    np.testing.assert_allclose(lg, z["prefill_logits"], atol=TOL, rtol=0)
    c.reset()
    T2 = max(1, meta["T"] // 2)
    lg = m.forward(c, z["prompt"][:T2], 0)
    np.testing.assert_allclose(lg, z["prefill_half_logits"], atol=TOL, rtol=0)


def test_greedy_tokens_and_step_logits(case):
    z, meta, cfg, m = case
    if not meta["n_gen"]:
        pytest.skip("prefill-only fixture")
    c = m.new_cache(64)
    toks, lg = m.generate(c, z["prompt"], meta["n_gen"], eos=-1, pos_mode="tokens", want_logits=True)
    np.testing.assert_array_equal(toks, z["gen_tokens"])           # bit-exact token ids
    np.testing.assert_allclose(lg, z["gen_logits"], atol=TOL, rtol=0)


def test_reference_position_mode(case):
    """Quirk C.1: Mistral/Qwen rotate decode call n as position n (mistral.rs:226,234)."""
    z, meta, cfg, m = case
    if not meta["ref_calls"]:
        pytest.skip("no reference-position vectors for this fixture")
    c = m.new_cache(64)
    toks, lg = m.generate(c, z["prompt"], meta["ref_calls"], eos=-1, pos_mode="reference", want_logits=True)
    np.testing.assert_array_equal(toks, z["ref_tokens"])
    np.testing.assert_allclose(lg, z["ref_logits"], atol=TOL, rtol=0)
    # and the two position modes really differ
    assert not np.allclose(z["ref_logits"][2], z["gen_logits"][2], atol=1e-3)


def test_bf16_weight_storage_matches_f32(case):
    z, meta, cfg, m = case
    w = synth.synth_weights(cfg)
    mb = oracle.OracleModel(cfg, w)           # uint16 = bf16 bits, kept as bf16
    c1, c2 = m.new_cache(32), mb.new_cache(32)
    np.testing.assert_array_equal(m.forward(c1, z["prompt"], 0), mb.forward(c2, z["prompt"], 0))


def _teacher_forced(model, z, meta, cap=64):
    """Last-position logits of the prefill and of every decode step, fed the fixture's fp32 greedy tokens."""
    c = model.new_cache(cap)
    out = [model.forward(c, z["prompt"], 0)]
    if meta["n_gen"]:
        for i, tok in enumerate(z["gen_tokens"][:-1]):
            out.append(model.forward(c, [int(tok)], meta["T"] + i))
    return np.stack(out)


def _fp32_rows(z, meta):
    return np.concatenate([z["prefill_logits"][None], z["gen_logits"][1:]]) if meta["n_gen"] else z["prefill_logits"][None]


def test_candle_bf16_emulation_sits_where_a_bf16_execution_sits(case):
    """round_bf16 = 2 restates candle's bf16 execution (every op's output a bf16 tensor).  There is no
    candle here to pin it to; the nearest independent datum is HuggingFace run in bfloat16 on the same weights
    (hf_bf16_* in the fixtures).  Both are "round after every op" executions of the same architecture, so their
    distances to the fp32 logits must be of the same size -- and the emulation must not be the fp32 or the
    product-rounding mode in disguise."""
    z, meta, cfg, m = case
    w = synth.as_f32(synth.synth_weights(cfg))
    ref = _fp32_rows(z, meta)
    hf = np.concatenate([z["hf_bf16_prefill_logits"][None], z["hf_bf16_gen_logits"][1:]]) if meta["n_gen"] else z["hf_bf16_prefill_logits"][None]
    cand = _teacher_forced(oracle.OracleModel(cfg, w, round_bf16=2), z, meta)
    prod = _teacher_forced(oracle.OracleModel(cfg, w, round_bf16=1), z, meta)
    n = np.linalg.norm(ref)
    e_hf, e_cand, e_prod = (np.linalg.norm(a - ref) / n for a in (hf, cand, prod))
    assert 0.4 * e_hf <= e_cand <= 2.5 * e_hf, (e_hf, e_cand)
    assert e_prod < e_cand, (e_prod, e_cand)          # fp32 residual stream + fewer rounding points: closer to fp32
    # bf16 logits: every value of the emulation is representable in bf16
    assert np.array_equal(synth.bf16_bits_to_f32(synth.f32_to_bf16_bits(cand)), cand)
