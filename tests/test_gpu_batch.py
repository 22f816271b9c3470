"""Batched decode (row N4: concurrent streams share one read of the weights): fl_batch_* against the
single-sequence entry points on the same caches' contents, and against the oracle.

B <= 2 runs the VALU kernel, which does the arithmetic of the single-sequence kernels (same per-lane K
order and butterfly in the dot products; only the fp32 order of the norm's sum of squares may differ), so
its logits agree with the single-sequence path to ~1e-6.  B >= 3 runs the MFMA kernel: the matrix core sums
K in another order, bf16 rounding points downstream may flip, so it is held to the bf16 bar of the other
parity tests (against the oracle and against the single-sequence path).
"""
import numpy as np
import pytest

import synth
from conftest import needs_experimental
from oracle import oracle
from test_gpu_parity import check_logits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import fastllm_amd
    assert fastllm_amd.device_count() >= 1
    return fastllm_amd


def prefilled(gm, cfg, lens, cap=96, seed0=40):
    caches, firsts, prompts = [], [], []
    for i, n in enumerate(lens):
        p = synth.prompt_ids(cfg, n, seed=seed0 + i)
        c = gm.new_cache(cap)
        firsts.append(gm.forward_argmax(c, p, 0))
        caches.append(c)
        prompts.append(p)
    return caches, firsts, prompts


def tight(a, b, what, B=1):
    n = np.linalg.norm(b)
    lim = 2e-3 if B <= 2 else 1e-2
    assert np.linalg.norm(a - b) <= lim * n, "%s: rel L2 %.2e" % (what, np.linalg.norm(a - b) / n)


@pytest.mark.parametrize("name", ["llama_a", "mistral_a", "qwen2_a", "llama_mha", "llama_tp4"])
@pytest.mark.parametrize("B", [1, 3, 8, 9, 16, 32])
def test_batch_forward_matches_single_and_oracle(fa, name, B):
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype="bf16")
    om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=True)
    lens = [5 + (3 * i) % 41 for i in range(B)]                # ragged: every sequence at its own position
    caches, firsts, prompts = prefilled(gm, cfg, lens)
    singles, _, _ = prefilled(gm, cfg, lens)
    ocs = []
    for p in prompts:
        oc = om.new_cache(96)
        om.forward(oc, p, 0)
        ocs.append(oc)
    batch = fa.Batch(gm, caches)
    toks = list(firsts)
    for step in range(4):
        pos = [lens[i] + step for i in range(B)]
        lg, am = batch.forward(toks, pos)
        for i in range(B):
            ref = gm.forward(singles[i], [toks[i]], pos[i])
            tight(lg[i], ref, "%s seq %d step %d vs single" % (name, i, step), B)
            check_logits(lg[i], om.forward(ocs[i], [toks[i]], pos[i]), "bf16", "%s seq %d step %d vs oracle" % (name, i, step))
            assert am[i] == oracle.argmax(lg[i])
            assert len(caches[i]) == lens[i] + step + 1
        toks = [int(t) for t in am]
    batch.close()


@pytest.mark.parametrize("B", [2, 5, 8, 16, 32, 64])
def test_batch_decode_loop_matches_single_loops(fa, B):
    cfg = synth.CONFIGS["mistral_a"]
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype="bf16")
    lens = [4 + (2 * i) % 37 for i in range(B)]
    caches, firsts, _ = prefilled(gm, cfg, lens)
    singles, firsts2, _ = prefilled(gm, cfg, lens)
    assert firsts == firsts2
    n = 20
    batch = fa.Batch(gm, caches)
    got = batch.decode(firsts, lens, n)
    for i in range(B):
        want = gm.decode_greedy(singles[i], firsts[i], lens[i], n)
        assert len(got[i]) == n and len(caches[i]) == lens[i] + n
        if not np.array_equal(got[i], want):                    # only a near-tie may separate the two paths
            k = int(np.argmax(got[i] != want))
            c = gm.new_cache(96)
            gm.forward(c, synth.prompt_ids(cfg, lens[i], seed=40 + i), 0)
            lg = None
            seq = [firsts[i]] + want[:k].tolist()
            for s, t in enumerate(seq):
                lg = gm.forward(c, [t], lens[i] + s)
            top2 = np.sort(lg)[-2:]
            assert top2[1] - top2[0] < 4e-2 * max(1.0, np.abs(lg).max()), "seq %d diverges at %d with a clear margin" % (i, k)
    # a second call continues from where the first stopped (graph replay, state re-set per call)
    more = batch.decode([int(g[-1]) for g in got], [lens[i] + n for i in range(B)], 5)
    assert all(len(m) == 5 for m in more)
    batch.close()


def test_batch_eos_and_sampling(fa):
    cfg = synth.CONFIGS["llama_a"]
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype="bf16")
    lens = [6, 9, 7, 11]
    caches, firsts, _ = prefilled(gm, cfg, lens)
    batch = fa.Batch(gm, caches)
    full = batch.decode(firsts, lens, 16)
    eos = int(full[1][5])                                        # make sequence 1 stop early
    caches2, firsts2, _ = prefilled(gm, cfg, lens)
    b2 = fa.Batch(gm, caches2)
    got = b2.decode(firsts2, lens, 16, eos=eos)
    for i in range(4):
        np.testing.assert_array_equal(got[i], full[i][: len(got[i])])       # the same batch path with and without EOS
        stop = np.flatnonzero(full[i] == eos)
        n_i = int(stop[0]) if len(stop) else 16
        assert len(got[i]) == n_i                                # the EOS token itself is not emitted (mod.rs:431-436)
        assert len(caches2[i]) == lens[i] + min(n_i + 1, 16)     # same cache length rule as fl_decode_greedy
    assert len(got[1]) <= 5
    # temperature sampling: every sequence draws from its own seed-0 stream, like separate requests
    caches3, firsts3, _ = prefilled(gm, cfg, lens)
    singles3, _, _ = prefilled(gm, cfg, lens)
    b3 = fa.Batch(gm, caches3)
    sam = b3.decode(firsts3, lens, 12, temperature=0.8, draws_done=0)
    assert all(len(x) == 12 for x in sam) and any(not np.array_equal(sam[i], full[i][:12]) for i in range(4))
    # two sequences (VALU kernel: the single-sequence arithmetic): token for token the single-stream sampler
    b4 = fa.Batch(gm, singles3[:2])
    two = b4.decode(firsts3[:2], lens[:2], 12, temperature=0.8, draws_done=0)
    ref, _, _ = prefilled(gm, cfg, lens[:2])
    for i in range(2):
        np.testing.assert_array_equal(two[i], gm.decode_sample(ref[i], firsts3[i], lens[i], 12, 0.8, draws_done=0))
    b4.close()
    for b in (batch, b2, b3):
        b.close()


def test_batch_k_slices(fa):
    """An intermediate size whose B = 8 activation rows do not fit LDS at once: down_proj runs as K slices
    that write separate slabs, summed by the next norm prologue."""
    cfg = dict(family="llama", hidden_size=256, intermediate_size=9728, vocab_size=256, num_hidden_layers=2,
               num_attention_heads=4, num_key_value_heads=2, rms_norm_eps=1e-5, rope_theta=10000.0, max_position_embeddings=512)
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype="bf16")
    lens = [3 + i for i in range(8)]
    caches, firsts, _ = prefilled(gm, cfg, lens)
    singles, _, _ = prefilled(gm, cfg, lens)
    batch = fa.Batch(gm, caches)
    lg, am = batch.forward(firsts, lens)
    for i in range(8):
        tight(lg[i], gm.forward(singles[i], [firsts[i]], lens[i]), "k-sliced down_proj, seq %d" % i, 8)
    batch.close()


@pytest.mark.parametrize("name,B", [("llama_a", 3), ("qwen2_a", 5), ("mistral_win", 9), ("llama_d100", 2)])
def test_batch_fp32_matches_single_and_oracle(fa, name, B):
    """Round 5: batches of an fp32 model (the literal-parity mode) -- one pass over the weights per step for all B rows; embedding, RoPE /
    KV append and attention as the single-sequence kernels on row i.  Logits within the fp32 bar (1e-3) of the oracle at every step,
    and of the single-sequence path (another order of the fp32 sums in the projections)."""
    from test_gpu_parity import FP32_TOL
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype="f32")
    om = oracle.OracleModel(cfg, synth.as_f32(w))
    lens = [4 + (5 * i) % 23 for i in range(B)]
    caches, firsts, prompts = prefilled(gm, cfg, lens)
    singles, _, _ = prefilled(gm, cfg, lens)
    ocs = []
    for p in prompts:
        oc = om.new_cache(96)
        om.forward(oc, p, 0)
        ocs.append(oc)
    batch = fa.Batch(gm, caches)
    toks = list(firsts)
    for step in range(5):
        pos = [lens[i] + step for i in range(B)]
        lg, am = batch.forward(toks, pos)
        for i in range(B):
            ref = gm.forward(singles[i], [toks[i]], pos[i])
            np.testing.assert_allclose(lg[i], ref, atol=2e-4, rtol=0, err_msg="%s seq %d step %d vs single" % (name, i, step))
            np.testing.assert_allclose(lg[i], om.forward(ocs[i], [toks[i]], pos[i]), atol=FP32_TOL, rtol=0, err_msg="%s seq %d step %d vs oracle" % (name, i, step))
            assert am[i] == oracle.argmax(lg[i])
            assert len(caches[i]) == lens[i] + step + 1
        toks = [int(t) for t in am]
    # ... and the multi-step entry point (graph replay of the 3 B + 6 launches per layer)
    out = batch.decode(toks, [lens[i] + 5 for i in range(B)], 6)
    for i in range(B):
        ref = gm.decode_greedy(singles[i], toks[i], lens[i] + 5, 6)
        assert [int(t) for t in out[i]] == [int(t) for t in ref], "%s seq %d greedy tokens" % (name, i)
    batch.close()


@pytest.mark.parametrize("mode", ["plain", "mixed"])
def test_batch_without_the_mfma_attention_layout(fa, mode):
    """bf16 caches in the plain layout (FL_ATTN_MFMA=0 stands in for a head shape the MFMA attention does not take): the batch runs
    the plain-layout batch kernels -- or, when the caches of one batch differ in layout ("mixed"), RoPE / attention per sequence --
    instead of being refused."""
    cfg = synth.CONFIGS["mistral_a"]
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype="bf16")
    om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=True)
    B, lens = 4, [7, 12, 5, 20]
    try:
        fa.tune("attn_mfma", 0)
        caches, firsts, prompts = prefilled(gm, cfg, lens)
        singles, _, _ = prefilled(gm, cfg, lens)
    finally:
        fa.tune("reload_env", 0)
    if mode == "mixed":                                    # sequences 1 and 3 on caches in the MFMA layout
        c2, f2, _ = prefilled(gm, cfg, lens)
        for i in (1, 3):
            caches[i].close()
            caches[i] = c2[i]
            assert f2[i] == firsts[i]
    ocs = []
    for p in prompts:
        oc = om.new_cache(96)
        om.forward(oc, p, 0)
        ocs.append(oc)
    batch = fa.Batch(gm, caches)
    toks = list(firsts)
    for step in range(3):
        pos = [lens[i] + step for i in range(B)]
        lg, am = batch.forward(toks, pos)
        for i in range(B):
            tight(lg[i], gm.forward(singles[i], [toks[i]], pos[i]), "plain layout seq %d step %d vs single" % (i, step), B)
            check_logits(lg[i], om.forward(ocs[i], [toks[i]], pos[i]), "bf16", "plain layout seq %d step %d vs oracle" % (i, step))
        toks = [int(t) for t in am]
    batch.close()


@pytest.mark.parametrize("name,dtype", [("mistral_a", "bf16"), ("qwen2_a", "bf16"), ("llama_a", "f32")])
def test_continuous_batching_replace_a_sequence(fa, name, dtype):
    """fl_batch_replace (mod.rs:137-238: streams come and go independently): after some steps one stream leaves and
    another -- other prompt, other length, a larger cache (more attention splits than the batch had: the graph is re-captured) --
    takes its slot, the batch goes on.  Every stream's greedy tokens equal the single-sequence path's on a twin cache, before and
    after the swap; the cache that left continues on its own."""
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype=dtype)
    B, lens = 5, [6, 11, 4, 9, 15]
    caches, firsts, _ = prefilled(gm, cfg, lens)
    singles, _, _ = prefilled(gm, cfg, lens)
    batch = fa.Batch(gm, caches)
    out1 = batch.decode(firsts, lens, 7)
    ref1 = [gm.decode_greedy(singles[i], firsts[i], lens[i], 7) for i in range(B)]
    for i in range(B):
        assert [int(t) for t in out1[i]] == [int(t) for t in ref1[i]], "seq %d before the swap" % i
    # stream 2 leaves; a new one joins in its slot (a cache eight times as large; a 13-token prompt)
    pn = synth.prompt_ids(cfg, 13, seed=977)
    cn, cn_twin = gm.new_cache(420), gm.new_cache(420)
    fn = gm.forward_argmax(cn, pn, 0)
    assert gm.forward_argmax(cn_twin, pn, 0) == fn
    left, left_twin = caches[2], singles[2]
    batch.replace(2, cn)
    first2 = [int(out1[i][-1]) for i in range(B)]
    pos2 = [lens[i] + 7 for i in range(B)]
    first2[2], pos2[2] = fn, 13
    out2 = batch.decode(first2, pos2, 9)
    for i in range(B):
        ref = gm.decode_greedy(cn_twin if i == 2 else singles[i], first2[i], pos2[i], 9)
        assert [int(t) for t in out2[i]] == [int(t) for t in ref], "seq %d after the swap" % i
    # the stream that left goes on alone where the batch left it
    a = gm.decode_greedy(left, int(out1[2][-1]), lens[2] + 7, 5)
    bb = gm.decode_greedy(left_twin, int(ref1[2][-1]), lens[2] + 7, 5)
    assert [int(t) for t in a] == [int(t) for t in bb]
    with pytest.raises(fa.FastLLMError):
        batch.replace(0, cn)                                     # already sequence 2 of this batch
    with pytest.raises(fa.FastLLMError):
        batch.replace(B, left)                                   # no such slot
    batch.replace(4, left)                                       # ... and a stream may come back
    batch.close()


def test_batch_errors(fa):
    cfg = synth.CONFIGS["llama_a"]
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype="bf16")
    c = gm.new_cache(16)
    with pytest.raises(fa.FastLLMError):
        fa.Batch(gm, [c, c])                                     # the same cache twice
    with pytest.raises(fa.FastLLMError):
        fa.Batch(gm, [gm.new_cache(8) for _ in range(65)])       # more than 64
    b = fa.Batch(gm, [c])
    with pytest.raises(fa.FastLLMError) as e:
        b.decode([1], [0], 17)                                   # beyond the cache capacity
    assert e.value.code == -7
    b.close()


def test_batch_fullsize_tinyllama(fa):
    """BASELINE shape (TinyLlama-1.1B), 8 streams: per-sequence logits equal the single-stream path's."""
    from fastllm_amd.configs import MODEL_CONFIGS
    from test_gpu_fullsize import pooled_weights, close_bf16
    cfg = MODEL_CONFIGS["tinyllama-1.1b"]
    w = pooled_weights(cfg)
    gm = fa.Model(cfg, w, dtype="bf16")
    lens = [16 + 5 * i for i in range(8)]
    caches, firsts, _ = prefilled(gm, cfg, lens, cap=128)
    singles, _, _ = prefilled(gm, cfg, lens, cap=128)
    batch = fa.Batch(gm, caches)
    lg, am = batch.forward(firsts, lens)
    for i in range(8):
        close_bf16(lg[i], gm.forward(singles[i], [firsts[i]], lens[i]), "tinyllama seq %d" % i)
    got = batch.decode([int(t) for t in am], [n + 1 for n in lens], 24)
    assert all(len(g) == 24 for g in got)
    batch.close()


@needs_experimental
@pytest.mark.parametrize("B", [3, 8, 16, 40])
def test_batch_five_launch_layer_matches_the_default_step(fa, B):
    """FL_GEMM_SKF=2: the decode batch's layer as five launches (k_gemm_skf.hip: every row's RoPE / KV append in the QKV epilogue,
    residual + norm in o_proj's and down_proj's, K slices met inside the launch) against the default eight-launch step on the same
    caches' contents; and 9-16 rows on the LDS-DMA ring kernel (k_gemv_dma.hip took eight at most before round 5)."""
    cfg = synth.CONFIGS["mistral_a"]
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype="bf16")
    lens = [4 + (3 * i) % 29 for i in range(B)]
    out = {}
    try:
        for mode in (2, 1):
            fa.tune("gemm_skf", mode)
            caches, firsts, _ = prefilled(gm, cfg, lens)
            batch = fa.Batch(gm, caches)
            lg, am = batch.forward(firsts, lens)
            lg2, am2 = batch.forward([int(t) for t in am], [n + 1 for n in lens])
            out[mode] = (lg, lg2, batch.decode([int(t) for t in am2], [n + 2 for n in lens], 12))     # (the graph-replayed loop)
            batch.close()
    finally:
        fa.tune("reload_env", 0)
    for i in range(B):
        tight(out[2][0][i], out[1][0][i], "five-launch layer, seq %d" % i, B)
        tight(out[2][1][i], out[1][1][i], "five-launch layer, second step, seq %d" % i, B)
        assert len(out[2][2][i]) == 12
    gm.close()
