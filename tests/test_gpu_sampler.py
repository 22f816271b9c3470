"""Device-side token selection (select_advance_kernel) against the oracle's LogitsProcessor.

The kernel keeps the reference's left-to-right fp32 order for the softmax denominator and the cumulative
weights (one lane walks them; at V ~ 1e5 any other order moves the result by several tokens in a flat
region) and evaluates exp in fp64-then-round.  The bar: the same token as the oracle on every draw; the
only tolerated difference is a draw whose `chosen` lies within MARGIN of a boundary (an exp that is not
correctly rounded on one side moves a cumulative weight by an ulp).
"""
import numpy as np
import pytest

import synth
from oracle import oracle

pytestmark = pytest.mark.gpu
MARGIN = 1e-6          # |chosen - boundary| (total ~ 1) below which a 1-ulp difference in an exp may flip the token


@pytest.fixture(scope="module")
def fa():
    import fastllm_amd
    assert fastllm_amd.device_count() >= 1
    return fastllm_amd


def compare_draws(got, lg, seed, temperature, draws_done=0):
    o = oracle.Sampler(seed, temperature)
    for _ in range(draws_done):
        o.next_u32()
    n_diff = 0
    for i, g in enumerate(got):
        tok, info = o.sample(lg, want_info=True)
        if int(g) == tok:
            continue
        n_diff += 1
        lo, hi = (info["cum_lo"], info["cum_hi"])
        near = min(abs(info["chosen"] - lo), abs(info["chosen"] - hi))
        assert near <= MARGIN, "draw %d: device %d, oracle %d, chosen %.9g in [%.9g, %.9g)" % (i, g, tok, info["chosen"], lo, hi)
    return n_diff


@pytest.mark.parametrize("V,temperature,seed", [(320, 0.8, 0), (32000, 1.0, 0), (32000, 0.3, 11), (152064, 0.7, 0),
                                                (151, 1.5, 3), (50257, 2.0, 123456789)])
def test_kernel_draws_match_oracle(fa, V, temperature, seed, monkeypatch):
    rs = np.random.RandomState(V + seed)
    lg = (rs.randn(V) * 2.5).astype(np.float32)
    n = 400
    got = fa.op_sample(lg, n, temperature, seed)
    assert got.max() < V
    n_diff = compare_draws(got, lg, seed, temperature)
    assert n_diff <= 1, n_diff
    # the parallel ordered sum is the one-lane walk, bit for bit
    monkeypatch.setenv("FL_SAMPLE_WALK", "1")
    np.testing.assert_array_equal(fa.op_sample(lg, n, temperature, seed), got)


@pytest.mark.parametrize("shape", ["flat", "peaked", "one_hot", "tiny_tail", "ties"])
def test_ordered_sum_equals_walk_on_hard_inputs(fa, shape, monkeypatch):
    """Distributions that stress the binade bookkeeping: totals that sit at a power of two, masses that
    vanish against the running sum, exact half-ulp ties (values that are small powers of two)."""
    V = 32768
    rs = np.random.RandomState(7)
    if shape == "flat":
        lg = np.zeros(V, np.float32)                      # p = 1/V exactly: every add is exact, total = 1.0
    elif shape == "peaked":
        lg = (rs.randn(V) * 0.5).astype(np.float32); lg[12345] = 40.0
    elif shape == "one_hot":
        lg = np.full(V, -200.0, np.float32); lg[777] = 0.0  # all other exps are 0 / denormal
    elif shape == "tiny_tail":
        lg = np.concatenate([np.full(64, 5.0), np.full(V - 64, -12.0)]).astype(np.float32)
    else:
        lg = (np.log(2.0) * rs.randint(-20, 1, size=V)).astype(np.float32)   # powers of two: ties everywhere
    a = fa.op_sample(lg, 300, 1.0, 1)
    n_diff = compare_draws(a, lg, 1, 1.0)
    assert n_diff <= 1, n_diff
    monkeypatch.setenv("FL_SAMPLE_WALK", "1")
    np.testing.assert_array_equal(fa.op_sample(lg, 300, 1.0, 1), a)


def test_kernel_stream_position(fa):
    """draws_done positions the stream: draws [5, 25) of one run equal a run started at word 5."""
    lg = (np.random.RandomState(1).randn(1000) * 3).astype(np.float32)
    a = fa.op_sample(lg, 25, 0.9, 4)
    b = fa.op_sample(lg, 20, 0.9, 4, draws_done=5)
    np.testing.assert_array_equal(a[5:], b)
    assert len(set(a.tolist())) > 5


def test_kernel_argmax_below_threshold(fa):
    lg = np.zeros(5000, dtype=np.float32)
    lg[[17, 4321]] = 3.0                              # tie: last maximal index
    assert fa.op_sample(lg, 3, 0.0).tolist() == [4321] * 3
    assert fa.op_sample(lg, 3, 9e-8).tolist() == [4321] * 3


def test_kernel_distribution(fa):
    rs = np.random.RandomState(5)
    lg = (rs.randn(64) * 1.5).astype(np.float32)
    n = 50000
    got = fa.op_sample(lg, n, 0.9, 2)
    p = np.exp(lg.astype(np.float64) / 0.9)
    p /= p.sum()
    counts = np.bincount(got, minlength=64)
    chi2 = ((counts - n * p) ** 2 / (n * p)).sum()
    assert chi2 < 130, chi2                           # 63 dof: 99.9th percentile ~ 103


@pytest.mark.parametrize("name,dtype", [("llama_a", "f32"), ("qwen2_a", "bf16")])
def test_generate_with_temperature_matches_oracle_loop(fa, name, dtype):
    """The reference's request shape (mod.rs:363-463) with temperature > 0: a fresh seed-0 processor, first
    token from the prefill logits, then the decode loop -- here fl_forward_sample + fl_decode_sample (device
    loop, graph-replayed) against forward-on-oracle + oracle sampler, teacher-forced on the device's tokens."""
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype=dtype)
    om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=(dtype == "bf16"))
    ids = synth.prompt_ids(cfg, 9, seed=2)
    T, n, temp = len(ids), 24, 0.8
    gc = gm.new_cache(64)
    first = gm.forward_sample(gc, ids, 0, temp)
    rest = gm.decode_sample(gc, first, T, n - 1, temp, draws_done=1)
    toks = [first] + rest.tolist()
    assert len(toks) == n and len(gc) == T + n - 1
    # the device's own logits along its own token sequence, sampled by the oracle's processor
    c2 = gm.new_cache(64)
    o = oracle.Sampler(0, temp)
    lg = gm.forward(c2, ids, 0)
    for i in range(n):
        tok, info = o.sample(lg, want_info=True)
        if tok != toks[i]:
            near = min(abs(info["chosen"] - info["cum_lo"]), abs(info["chosen"] - info["cum_hi"]))
            assert near <= MARGIN, "step %d: device %d oracle %d" % (i, toks[i], tok)
        if i + 1 < n:
            lg = gm.forward(c2, [toks[i]], T + i)
    # and the whole loop on the oracle model: same tokens as long as the bf16/f32 logits agree closely
    oc = om.new_cache(64)
    o2 = oracle.Sampler(0, temp)
    lg = om.forward(oc, ids, 0)
    same = 0
    for i in range(n):
        if o2.sample(lg) != toks[i]:
            break
        same += 1
        lg = om.forward(oc, [toks[i]], T + i)
    if dtype == "f32":
        assert same >= n - 1, same
    # a greedy call afterwards is ArgMax again (the selection state belongs to the call, not the cache)
    c3 = gm.new_cache(64)
    assert gm.forward_argmax(c3, ids, 0) == oracle.argmax(gm.forward(gm.new_cache(64), ids, 0))
