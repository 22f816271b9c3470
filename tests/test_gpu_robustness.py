"""GPU tests of the host-side robustness fixes of round 2: the exception barrier on entry points that
need a device, the EOS-bounded decode loop, and the prefill scratch that is replaced instead of piling up."""
import time

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import fastllm_amd
    assert fastllm_amd.device_count() >= 1, "no MI355X visible"
    return fastllm_amd


def test_exception_barrier_in_cache_create_and_forward(fa, monkeypatch):
    cfg = synth.CONFIGS["llama_a"]
    m = fa.Model(cfg, synth.synth_weights(cfg), dtype="f32")
    monkeypatch.setenv("FL_DEBUG_THROW", "cache_create=bad_alloc")
    with pytest.raises(fa.FastLLMError) as e:
        m.new_cache(32)
    assert e.value.code == -4                                # FL_ERR_OOM
    monkeypatch.delenv("FL_DEBUG_THROW")
    c = m.new_cache(32)
    ids = synth.prompt_ids(cfg, 8)
    monkeypatch.setenv("FL_DEBUG_THROW", "forward=runtime")
    with pytest.raises(fa.FastLLMError) as e:
        m.forward(c, ids, 0)
    assert e.value.code == -5 and "injected failure" in str(e.value)
    monkeypatch.delenv("FL_DEBUG_THROW")
    assert len(c) == 0                                       # the failed call left the cache untouched
    lg = m.forward(c, ids, 0)                                # and the model still works
    assert np.isfinite(lg).all() and len(c) == 8


def test_decode_with_eos_does_not_run_all_steps(fa):
    """mod.rs:431-436: the reference's loop breaks at the first EOS.  fl_decode_greedy enqueues 16, 32, ... steps between
    looks at the tokens, so an early EOS costs at most one chunk, not n_steps forwards."""
    cfg = dict(synth.CONFIGS["llama_a"], max_position_embeddings=4096)
    m = fa.Model(cfg, synth.synth_weights(cfg), dtype="f32")
    ids = synth.prompt_ids(cfg, 8)
    c = m.new_cache(2200)
    f = m.forward_argmax(c, ids, 0)
    full = m.decode_greedy(c, f, 8, 40)
    eos = int(full[5])
    hit = int(np.argmax(full == eos))
    c2 = m.new_cache(2200)
    f2 = m.forward_argmax(c2, ids, 0)
    assert f2 == f
    m.synchronize()
    t0 = time.perf_counter()
    got = m.decode_greedy(c2, f, 8, 2048, eos=eos)
    t_eos = time.perf_counter() - t0
    np.testing.assert_array_equal(got, full[:hit])
    assert len(c2) == 8 + hit + 1                           # the forward that produced EOS ran; nothing after it counts
    c3 = m.new_cache(2200)
    m.forward_argmax(c3, ids, 0)
    t0 = time.perf_counter()
    m.decode_greedy(c3, f, 8, 2048)
    t_all = time.perf_counter() - t0
    assert t_eos < 0.25 * t_all, (t_eos, t_all)             # 16 of 2048 steps (+ fixed costs), not all of them
    # a later EOS crosses chunk boundaries (16 + 32 = 48 steps): same tokens as the unbounded run
    late = m.decode_greedy(m_cache_after_prefill(m, ids), f, 8, 200, eos=-1)
    eos2 = int(late[70]) if int(late[70]) not in late[:70] else None
    if eos2 is not None:
        got2 = m.decode_greedy(m_cache_after_prefill(m, ids), f, 8, 200, eos=eos2)
        np.testing.assert_array_equal(got2, late[:70])


def m_cache_after_prefill(m, ids):
    c = m.new_cache(2200)
    m.forward_argmax(c, ids, 0)
    return c


def test_prefill_scratch_is_replaced_not_accumulated(fa):
    cfg = dict(synth.CONFIGS["llama_a"], max_position_embeddings=1024)
    w = synth.synth_weights(cfg)
    grown = fa.Model(cfg, w, dtype="bf16")
    ref_logits = {}
    for T in (40, 200, 600):                                  # three growing prompts: 128 -> 256 -> 640 rows of scratch
        c = grown.new_cache(T + 8)
        ref_logits[T] = grown.forward(c, synth.prompt_ids(cfg, T, seed=T), 0)
        c.close()
    direct = fa.Model(cfg, w, dtype="bf16")
    c = direct.new_cache(608)
    lg = direct.forward(c, synth.prompt_ids(cfg, 600, seed=600), 0)
    c.close()
    np.testing.assert_array_equal(lg, ref_logits[600])
    assert grown.info().hbm_bytes_allocated == direct.info().hbm_bytes_allocated
    # a shorter prompt afterwards reuses the set
    before = grown.info().hbm_bytes_allocated
    c = grown.new_cache(64)
    np.testing.assert_array_equal(grown.forward(c, synth.prompt_ids(cfg, 40, seed=40), 0), ref_logits[40])
    assert grown.info().hbm_bytes_allocated == before


def _against_oracle(fa, name, dtype, kw, T=10, n_dec=3):
    from oracle import oracle
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    ids = synth.prompt_ids(cfg, T + n_dec, seed=11)
    om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=(dtype == "bf16"))
    oc = om.new_cache(64)
    g = fa.Model(cfg, w, dtype=dtype, **kw)
    c = g.new_cache(64)
    tol = 1e-3 if dtype == "f32" else 0.3
    for i, (chunk, pos) in enumerate([(ids[:T], 0)] + [(ids[T + k:T + k + 1], T + k) for k in range(n_dec)]):
        a, o = g.forward(c, chunk, pos), om.forward(oc, chunk, pos)
        assert np.isfinite(a).all() and float(np.abs(a - o).max()) < tol, "%s %s %s call %d: max |diff| %.3g" % (name, dtype, kw, i, float(np.abs(a - o).max()))
    f = g.forward_argmax(c, ids[:1], T + n_dec)
    assert len(g.decode_greedy(c, f, T + n_dec + 1, 6)) == 6
    c.close()
    g.close()


def test_models_created_after_a_tensor_parallel_model_was_destroyed(fa):
    """Round 5, found by a flaky test: an FL_TP_SINGLE_PROCESS model's inbox is hipDeviceMallocUncached memory, and after its hipFree
    EVERY later model of the process -- single-GPU ones too -- computed garbage, deterministically (tools/seq_probe.py: 10 of 12
    wrong; with the inbox never handed back to the runtime, 0 of 36).  The library now keeps freed inboxes in a free list
    (comm.hip, inbox_acquire).  Models of every placement created and destroyed one after another, each against the oracle."""
    import gc
    from fastllm_amd import binding
    gc.collect()
    single = dict(tp_mode=binding.TP_SINGLE_PROCESS, tp_size=2, device_ids=[0, 0])
    emulated = dict(tp_mode=binding.TP_EMULATED, tp_size=2)
    for rnd in range(3):
        for name, dtype in (("llama_a", "bf16"), ("qwen2_a", "f32"), ("mistral_a", "bf16")):
            for kw in (single, {}, emulated, single):
                _against_oracle(fa, name, dtype, kw)


def test_results_do_not_depend_on_what_the_allocator_hands_back(fa, monkeypatch):
    """FL_DEBUG_POISON=255: every device allocation of a model / cache / batch is filled with 0xFF bytes (bf16 and fp32 NaN patterns)
    before use.  A kernel that reads bytes nobody wrote -- and gets away with it on fresh, zeroed HBM -- shows as a NaN here."""
    from fastllm_amd import binding
    monkeypatch.setenv("FL_DEBUG_POISON", "255")
    for name in ("llama_a", "qwen2_a", "mistral_win", "llama_mha", "mistral_d48", "llama_d100"):
        for dtype in ("bf16", "f32"):
            _against_oracle(fa, name, dtype, {})
            if (synth.CONFIGS[name].get("num_key_value_heads") or synth.CONFIGS[name]["num_attention_heads"]) % 2 == 0:
                _against_oracle(fa, name, dtype, dict(tp_mode=binding.TP_EMULATED, tp_size=2))
    # a mid-size prompt (in-launch K slices, their workspace) and a batch of streams on poisoned buffers
    cfg = synth.CONFIGS["mistral_a"]
    w = synth.synth_weights(cfg)
    g = fa.Model(cfg, w, dtype="bf16")
    c = g.new_cache(400)
    assert np.isfinite(g.forward(c, synth.prompt_ids(cfg, 300, seed=3), 0)).all()
    caches, firsts = [], []
    for i in range(12):
        ci = g.new_cache(64)
        firsts.append(g.forward_argmax(ci, synth.prompt_ids(cfg, 5 + i, seed=40 + i), 0))
        caches.append(ci)
    b = fa.Batch(g, caches)
    lg, am = b.forward(firsts, [5 + i for i in range(12)])
    assert np.isfinite(lg).all()
    b.close()
    g.close()
