"""Full-size checks on the BASELINE.json TinyLlama-1.1B shape (22 layers, h 2048, GQA 8, d 64, V 32000):
one direct oracle comparison plus size-independent properties of the HIP path (KV-cache equivalence,
fused/graph vs eager/unfused, MFMA vs VALU attention, TP-emulated vs TP=1, run-to-run determinism)."""
import zlib

import numpy as np
import pytest

import synth
from oracle import oracle
from test_gpu_parity import check_logits

pytestmark = pytest.mark.gpu


def pooled_weights(cfg, seed=7):
    """bf16 N(0, 0.02^2) weights cut from one 16M-value pool at a name-derived offset (fast to build)."""
    rs = np.random.RandomState(seed)
    pool = synth.f32_to_bf16_bits((0.02 * rs.standard_normal(1 << 24)).astype(np.float32))
    out = {}
    for name, shape in synth.tensor_shapes(cfg):
        n = int(np.prod(shape))
        off = zlib.crc32(name.encode()) % (1 << 23)
        reps = (off + n + len(pool) - 1) // len(pool)
        flat = np.tile(pool, reps)[off:off + n] if reps > 1 else pool[off:off + n]
        a = np.ascontiguousarray(flat.reshape(shape))
        if name.endswith("layernorm.weight") or name == "model.norm.weight":
            a = synth.f32_to_bf16_bits(1.0 + synth.bf16_bits_to_f32(a))
        elif name == "lm_head.weight":
            a = synth.f32_to_bf16_bits(8.0 * synth.bf16_bits_to_f32(a))      # decisive argmax
        out[name] = a
    return out


@pytest.fixture(scope="module")
def tiny():
    import fastllm_amd as fa
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = MODEL_CONFIGS["tinyllama-1.1b"]
    w = pooled_weights(cfg)
    return fa, cfg, w


def test_fullsize_vs_oracle(tiny):
    fa, cfg, w = tiny
    gm = fa.Model(cfg, w, dtype="bf16")
    om = oracle.OracleModel(cfg, w, round_bf16=True)          # bf16 weights kept as bf16 (exact)
    ids = synth.prompt_ids(cfg, 20, seed=2)
    gc, oc = gm.new_cache(64), om.new_cache(64)
    check_logits(gm.forward(gc, ids[:16], 0), om.forward(oc, ids[:16], 0), "bf16", "full-size prefill")
    for i in range(16, 20):
        check_logits(gm.forward(gc, ids[i:i + 1], i), om.forward(oc, ids[i:i + 1], i), "bf16", "full-size decode %d" % i)


def test_fullsize_kv_cache_equivalence_and_determinism(tiny):
    fa, cfg, w = tiny
    gm = fa.Model(cfg, w, dtype="bf16")
    ids = synth.prompt_ids(cfg, 128, seed=4)
    c1, c2 = gm.new_cache(256), gm.new_cache(256)
    full = gm.forward(c1, ids, 0)
    gm.forward(c2, ids[:96], 0)
    first = None
    for i in range(96, 128):
        part = gm.forward(c2, ids[i:i + 1], i)
    check_logits(part, full, "bf16", "prefill(128) vs prefill(96)+32 decodes")
    # greedy continuation twice from the same state: bit-identical ids (no atomics, fixed summation order)
    tok = int(np.argmax(full))
    a = gm.decode_greedy(c1, tok, 128, 48)
    c3 = gm.new_cache(256)
    gm.forward(c3, ids, 0)
    b = gm.decode_greedy(c3, tok, 128, 48)
    np.testing.assert_array_equal(a, b)
    assert len(a) == 48 and len(c1) == 128 + 48


@pytest.mark.parametrize("env", [{"FL_FUSED": "0", "FL_GRAPH": "0"}, {"FL_ATTN_MFMA": "0"}, {"FL_GRAPH": "0"}])
def test_fullsize_variant_paths_agree(tiny, env, monkeypatch):
    """fused + hipGraph + MFMA attention (default) vs the unfused / eager / VALU-attention variants."""
    fa, cfg, w = tiny
    ids = synth.prompt_ids(cfg, 40, seed=6)
    ref = fa.Model(cfg, w, dtype="bf16")
    rc = ref.new_cache(128)
    r0 = ref.forward(rc, ids[:32], 0)
    r1 = [ref.forward(rc, ids[i:i + 1], i) for i in range(32, 40)]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    alt = fa.Model(cfg, w, dtype="bf16")
    ac = alt.new_cache(128)
    check_logits(alt.forward(ac, ids[:32], 0), r0, "bf16", "prefill %s" % env)
    for i, want in zip(range(32, 40), r1):
        check_logits(alt.forward(ac, ids[i:i + 1], i), want, "bf16", "decode %d %s" % (i, env))


def test_fullsize_tp_emulated(tiny):
    from fastllm_amd import binding
    fa, cfg, w = tiny
    ids = synth.prompt_ids(cfg, 36, seed=8)
    g1 = fa.Model(cfg, w, dtype="bf16")
    c1 = g1.new_cache(64)
    a0 = g1.forward(c1, ids[:32], 0)
    a1 = [g1.forward(c1, ids[i:i + 1], i) for i in range(32, 36)]
    for tp in (2, 4):                               # Hkv = 4: TP <= 4 for TinyLlama
        gN = fa.Model(cfg, w, dtype="bf16", tp_mode=binding.TP_EMULATED, tp_size=tp)
        cN = gN.new_cache(64)
        check_logits(gN.forward(cN, ids[:32], 0), a0, "bf16", "tp%d prefill" % tp)
        for i, want in zip(range(32, 36), a1):
            check_logits(gN.forward(cN, ids[i:i + 1], i), want, "bf16", "tp%d decode" % tp)
        gN.close()
