"""The persistent decode engine (k_engine.hip, FL_ENGINE=1): one launch chains o_proj -> gate/up -> down_proj -> the next layer's
QKV projection (last layer: lm_head) with the activations crossing the chip as tagged granules.  Checked against the oracle
(same bands as the launch-per-projection path), against that path, for determinism and through the captured graph.
Path: the T = 1 forward of llama.rs:147-149 / mistral.rs:223-226 / qwen.rs:142."""
import numpy as np
import pytest

import synth
from oracle import oracle
from test_gpu_fullsize import close_bf16, pooled_weights
from test_gpu_parity import check_logits

from conftest import needs_experimental

pytestmark = [pytest.mark.gpu, needs_experimental]


@pytest.fixture()
def fa(monkeypatch):
    import fastllm_amd
    monkeypatch.setenv("FL_ENGINE", "1")
    return fastllm_amd


def engine_ran(m, c, tok, pos):
    m.profile_begin()
    m.forward(c, [tok], pos)
    return any(s["name"].startswith("gemv[eng") for s in m.profile_end())


@pytest.mark.parametrize("name", ["llama_a", "llama_mha", "mistral_a", "qwen2_a", "qwen2_win", "llama_tp4"])
def test_engine_decode_vs_oracle(fa, name):
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype="bf16")
    om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=True)
    T = 16
    ids = synth.prompt_ids(cfg, T + 20, seed=5)
    gc, oc = gm.new_cache(64), om.new_cache(64)
    check_logits(gm.forward(gc, ids[:T], 0), om.forward(oc, ids[:T], 0), "bf16", "prefill")
    for i in range(T, T + 19):                  # eager step, then the captured graph
        check_logits(gm.forward(gc, ids[i:i + 1], i), om.forward(oc, ids[i:i + 1], i), "bf16", "%s engine decode %d" % (name, i))
    assert engine_ran(gm, gc, int(ids[T + 19]), T + 19), "the engine launch did not run"
    gm.close()


def test_engine_greedy_loop_is_deterministic_and_matches_the_launch_path(fa, monkeypatch):
    cfg = synth.CONFIGS["mistral_a"]
    w = synth.synth_weights(cfg)
    ids = synth.prompt_ids(cfg, 12, seed=9)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FL_ENGINE", mode)
        gm = fa.Model(cfg, w, dtype="bf16")
        runs = []
        for _ in range(2):
            c = gm.new_cache(128)
            first = gm.forward_argmax(c, ids, 0)
            runs.append(np.concatenate([[first], gm.decode_greedy(c, first, len(ids), 80)]))
        np.testing.assert_array_equal(runs[0], runs[1])          # fixed summation order, no atomics: run-to-run identical
        c = gm.new_cache(128)
        gm.forward(c, ids, 0)
        out[mode] = (runs[0], [gm.forward(c, [int(t)], len(ids) + i) for i, t in enumerate(runs[0][:12])])
        gm.close()
    # the two paths sum a row's squares and the rows' dot products in the same order except the RMSNorm's sum of squares:
    # logits agree to bf16 noise, ids wherever the margin exceeds it
    for a, b in zip(out["1"][1], out["0"][1]):
        check_logits(a, b, "bf16", "engine vs launches")


def test_engine_tinyllama_full_size_vs_launch_path(fa, monkeypatch):
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = MODEL_CONFIGS["tinyllama-1.1b"]
    w = pooled_weights(cfg)
    ids = synth.prompt_ids(cfg, 40, seed=6)
    ref = None
    for mode in ("0", "1"):
        monkeypatch.setenv("FL_ENGINE", mode)
        gm = fa.Model(cfg, w, dtype="bf16")
        c = gm.new_cache(128)
        gm.forward(c, ids[:32], 0)
        got = [gm.forward(c, ids[i:i + 1], i) for i in range(32, 40)]
        if mode == "1":
            assert engine_ran(gm, c, 7, 40)
            for i, (a, b) in enumerate(zip(got, ref)):
                close_bf16(a, b, "TinyLlama decode %d, engine vs launches" % i)
        ref = got
        gm.close()


def test_engine_timeout_is_an_error_not_a_hang(fa, monkeypatch):
    """A grid that cannot be resident at once (here: forced by a grid twice the chip through fl_tune) must end in
    FL_ERR_HIP within the bound, never in a hung GPU.  Skipped where the tuning hook is absent."""
    import fastllm_amd.binding as binding
    cfg = synth.CONFIGS["llama_a"]
    w = synth.synth_weights(cfg)
    monkeypatch.setenv("FL_ENGINE_TIMEOUT_MS", "200")
    try:
        fa.tune("engine_grid", 512)
    except Exception:
        pytest.skip("no engine_grid tuning hook")
    try:
        gm = fa.Model(cfg, w, dtype="bf16")
        c = gm.new_cache(64)
        tok = gm.forward_argmax(c, synth.prompt_ids(cfg, 8), 0)
        with pytest.raises(binding.FastLLMError):
            gm.forward(c, [tok], 8)
        gm.close()
    finally:
        fa.tune("engine_grid", 0)
