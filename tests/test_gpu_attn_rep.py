"""Decode attention replicated inside the o_proj launch (k_attn_rep.hip): used for caches of up to ~96 KB of K / V per layer.
The small-cache tests of the whole suite run through it (against the oracle); here: that it really is the launch that runs, where
it stops being used, and that it agrees with the two launches it replaces at TinyLlama's width."""
import numpy as np
import pytest

import synth
from oracle import oracle
from test_gpu_fullsize import close_bf16, pooled_weights
from test_gpu_parity import check_logits

pytestmark = pytest.mark.gpu


def kernels_of_a_step(m, c, tok, pos):
    m.profile_begin()
    m.forward(c, [tok], pos)
    return [s["name"] for s in m.profile_end()]


@pytest.mark.parametrize("name", ["llama_a", "mistral_a", "qwen2_a", "llama_tp4"])
def test_replicated_attention_runs_and_matches_the_oracle(name):
    import fastllm_amd as fa
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype="bf16")
    om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=True)
    ids = synth.prompt_ids(cfg, 60, seed=12)
    gc, oc = gm.new_cache(64), om.new_cache(64)
    check_logits(gm.forward(gc, ids[:33], 0), om.forward(oc, ids[:33], 0), "bf16", "prefill")
    for i in range(33, 59):                                  # crosses a 32-key tile boundary: one and two tiles per wave
        check_logits(gm.forward(gc, ids[i:i + 1], i), om.forward(oc, ids[i:i + 1], i), "bf16", "%s decode %d" % (name, i))
    names = kernels_of_a_step(gm, gc, int(ids[59]), 59)
    assert any(n.startswith("attn_oproj[rep") for n in names) and not any(n.startswith("attn_decode") for n in names), names
    gm.close()


def test_tinyllama_width_short_cache_vs_two_launches_and_capacity_rule(monkeypatch):
    import fastllm_amd as fa
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS["tinyllama-1.1b"], num_hidden_layers=4)
    w = pooled_weights(cfg)
    ids = synth.prompt_ids(cfg, 90, seed=3)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FL_ATTN_REP", mode)
        gm = fa.Model(cfg, w, dtype="bf16")
        c = gm.new_cache(96)
        gm.forward(c, ids[:70], 0)
        out[mode] = [gm.forward(c, ids[i:i + 1], i) for i in range(70, 89)]
        names = kernels_of_a_step(gm, c, int(ids[89]), 89)
        assert any(n.startswith("attn_oproj[rep") for n in names) == (mode == "1"), names
        if mode == "1":                                      # a cache of 128 positions is beyond the rule: the two launches
            c2 = gm.new_cache(128)
            gm.forward(c2, ids[:16], 0)
            names2 = kernels_of_a_step(gm, c2, 5, 16)
            assert any(n.startswith("attn_decode") for n in names2) and not any(n.startswith("attn_oproj") for n in names2), names2
        gm.close()
    for i, (a, b) in enumerate(zip(out["1"], out["0"])):
        close_bf16(a, b, "decode %d, replicated vs two launches" % i)


def test_replicated_attention_in_an_emulated_tensor_parallel_group():
    import fastllm_amd as fa
    from fastllm_amd import binding
    cfg = synth.CONFIGS["llama_tp4"]
    w = synth.synth_weights(cfg)
    om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=True)
    ids = synth.prompt_ids(cfg, 40, seed=2)
    for tp in (2, 4):
        gm = fa.Model(cfg, w, dtype="bf16", tp_mode=binding.TP_EMULATED, tp_size=tp)
        gc, oc = gm.new_cache(64), om.new_cache(64)
        gm.forward(gc, ids[:20], 0); om.forward(oc, ids[:20], 0)
        for i in range(20, 36):
            check_logits(gm.forward(gc, ids[i:i + 1], i), om.forward(oc, ids[i:i + 1], i), "bf16", "tp%d decode %d" % (tp, i))
        assert any(n.startswith("attn_oproj[rep") for n in kernels_of_a_step(gm, gc, 3, 36))
        gm.close()
