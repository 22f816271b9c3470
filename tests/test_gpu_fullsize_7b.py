"""BASELINE.json's 7B shapes (configs C3-C5: Mistral-7B-v0.1, Qwen2-7B) on the GPU.

* full WIDTH against the oracle: the real h / heads / intermediate / vocabulary (incl. Qwen2's q/k/v
  bias and V = 152064) at 2 layers, fp32 within 1e-3 and bf16 against the bf16-emulating oracle;
* full DEPTH through size-independent properties (no CPU oracle could run 7B in test time): KV-cache
  equivalence, library-chunked prefill == one prefill (with Mistral's 4096 window crossed), emulated
  tensor parallelism == one GPU, run-to-run determinism of the greedy loop;
* Mistral's real window (4096) crossed by a 4300-token prompt on a narrow model, against the oracle.
Weights are generated in HBM with torch (as bench.py does); nothing here reads a checkpoint.
"""
import os
import sys

import numpy as np
import pytest

import synth
from conftest import needs_experimental
from oracle import oracle
from test_gpu_fullsize import close_bf16

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def env():
    import torch
    import fastllm_amd as fa
    sys.path.insert(0, ROOT)
    import bench
    assert torch.cuda.is_available()
    return torch, fa, bench


def host_copy(torch, wts):
    """device bf16 tensors -> numpy uint16 bf16 bits"""
    return {k: v.view(torch.int16).cpu().numpy().view(np.uint16) for k, v in wts.items()}


@pytest.mark.parametrize("name", ["mistral-7b", "qwen2-7b"])
def test_7b_full_width_two_layers_vs_oracle(env, name):
    torch, fa, bench = env
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS[name], num_hidden_layers=2)
    dev = torch.device("cuda", 0)
    wts = bench.synth_device_weights(torch, cfg, dev, seed=3)
    w = host_copy(torch, wts)
    del wts
    torch.cuda.empty_cache()
    g32, g16 = fa.Model(cfg, w, dtype="f32"), fa.Model(cfg, w, dtype="bf16")
    o32, oemu = oracle.OracleModel(cfg, w), oracle.OracleModel(cfg, w, round_bf16=True)
    ids = synth.prompt_ids(cfg, 28, seed=5)
    caches = [m.new_cache(64) for m in (g32, g16, o32, oemu)]

    def step(chunk, pos, what):
        a32, a16, r32, remu = [m.forward(c, chunk, pos) for m, c in zip((g32, g16, o32, oemu), caches)]
        np.testing.assert_allclose(a32, r32, atol=1e-3, rtol=0, err_msg="fp32 " + what)
        assert oracle.argmax(a32) == oracle.argmax(r32)
        n = np.linalg.norm(remu)
        assert np.linalg.norm(a16 - remu) <= 1e-2 * n, "%s: bf16 vs emulation rel L2 %.4f" % (what, np.linalg.norm(a16 - remu) / n)

    step(ids[:24], 0, name + " prefill")
    for i in range(24, 28):
        step(ids[i:i + 1], i, name + " decode %d" % i)
    for m in (g32, g16):
        m.close()


def test_mistral_7b_full_depth_properties(env, monkeypatch):
    torch, fa, bench = env
    from fastllm_amd import binding
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = MODEL_CONFIGS["mistral-7b"]
    dev = torch.device("cuda", 0)
    wts = bench.synth_device_weights(torch, cfg, dev, seed=9)
    tens = bench.as_fl_tensors(wts, 0)
    gm = fa.Model(cfg, tens, dtype="bf16")
    T = 512                                                     # config C3: 512-token prompt
    ids = synth.prompt_ids(cfg, T, seed=1)
    c1, c2 = gm.new_cache(1024), gm.new_cache(1024)
    full = gm.forward(c1, ids, 0)
    gm.forward(c2, ids[:T - 48], 0)
    for i in range(T - 48, T):
        part = gm.forward(c2, ids[i:i + 1], i)
    close_bf16(part, full, "mistral-7b prefill(512) vs prefill(464) + 48 decode steps")
    # greedy loop: deterministic, and the graph-replayed loop equals step-by-step forwards
    tok = int(np.argmax(full))
    a = gm.decode_greedy(c1, tok, T, 64)
    b = gm.decode_greedy(c2, tok, T, 64)
    assert len(a) == 64
    c3 = gm.new_cache(1024)
    gm.forward(c3, ids, 0)
    np.testing.assert_array_equal(gm.decode_greedy(c3, tok, T, 64), a)     # same path twice: bit-identical
    # c2 was built by another kernel path (bf16-close, not bit-equal): the two greedy sequences may part, but only
    # where the two candidates are a near-tie within that closeness -- replay cache 1's sequence up to the fork
    if (a != b).any():
        k = int(np.argmax(a != b))
        c5 = gm.new_cache(1024)
        lg = gm.forward(c5, ids, 0)
        seq = np.concatenate([[tok], a[:k]]).astype(np.uint32)
        for i, t_in in enumerate(seq):
            lg = gm.forward(c5, seq[i:i + 1], T + i)
        assert int(np.flatnonzero(lg == lg.max())[-1]) == int(a[k])
        margin = float(lg[a[k]] - lg[b[k]])
        assert 0.0 <= margin <= 6e-2 * max(1.0, float(np.abs(lg).max())), "fork at step %d is not a near-tie: margin %g" % (k, margin)
    # library-chunked prefill keeps the single call's mask
    monkeypatch.setenv("FL_PREFILL_CHUNK", "160")
    c4 = gm.new_cache(1024)
    close_bf16(gm.forward(c4, ids, 0), full, "mistral-7b prefill in 160-token chunks")
    monkeypatch.delenv("FL_PREFILL_CHUNK")
    # tensor parallelism (emulated on one GPU): TP=8 is config C4
    for tp in (2, 8):
        gN = fa.Model(cfg, tens, dtype="bf16", tp_mode=binding.TP_EMULATED, tp_size=tp)
        cN = gN.new_cache(1024)
        close_bf16(gN.forward(cN, ids, 0), full, "mistral-7b tp%d prefill" % tp)
        lg1, lgN = gm.forward(c4, [tok], T), gN.forward(cN, [tok], T)
        close_bf16(lgN, lg1, "mistral-7b tp%d decode" % tp)
        c4.reset()
        gm.forward(c4, ids, 0)
        gN.close()
    gm.close()


def test_qwen2_7b_full_depth_4k_prefill(env, monkeypatch):
    """Config C5: 4096-token prefill + decode at S = 4097 (here single GPU and emulated TP=4)."""
    torch, fa, bench = env
    from fastllm_amd import binding
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = MODEL_CONFIGS["qwen2-7b"]
    dev = torch.device("cuda", 0)
    wts = bench.synth_device_weights(torch, cfg, dev, seed=10)
    tens = bench.as_fl_tensors(wts, 0)
    gm = fa.Model(cfg, tens, dtype="bf16")
    T = 4096
    ids = synth.prompt_ids(cfg, T, seed=3)
    c1 = gm.new_cache(T + 64)
    full = gm.forward(c1, ids, 0)
    d1 = gm.forward(c1, [7], T)
    monkeypatch.setenv("FL_PREFILL_CHUNK", "1000")               # 4096 = 1000 x 4 + 96
    c2 = gm.new_cache(T + 64)
    close_bf16(gm.forward(c2, ids, 0), full, "qwen2-7b 4096-token prefill in 1000-token chunks")
    close_bf16(gm.forward(c2, [7], T), d1, "decode at S = 4097 after the chunked prefill")
    monkeypatch.delenv("FL_PREFILL_CHUNK")
    c3 = gm.new_cache(T + 64)
    gm.forward(c3, ids[:T - 6], 0)
    for i in range(T - 6, T):
        part = gm.forward(c3, ids[i:i + 1], i)
    close_bf16(part, full, "qwen2-7b prefill(4096) vs prefill(4090) + 6 decode steps")
    gN = fa.Model(cfg, tens, dtype="bf16", tp_mode=binding.TP_EMULATED, tp_size=4)
    cN = gN.new_cache(T + 64)
    close_bf16(gN.forward(cN, ids, 0), full, "qwen2-7b tp4 prefill")
    close_bf16(gN.forward(cN, [7], T), d1, "qwen2-7b tp4 decode at S = 4097")
    gN.close()
    gm.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_window_4096_crossed_vs_oracle(env, dtype):
    """Mistral's real sliding window: a 4300-token prompt on a narrow model (head_dim 128, GQA 2) against the
    oracle -- the prefill mask drops keys older than 4096 (+1, App. A.5), the decode step sees the whole cache."""
    torch, fa, bench = env
    cfg = dict(family="mistral", hidden_size=256, intermediate_size=512, vocab_size=256, num_hidden_layers=1,
               num_attention_heads=2, num_key_value_heads=1, rms_norm_eps=1e-5, rope_theta=10000.0,
               max_position_embeddings=8192, sliding_window=4096)
    w = synth.synth_weights(cfg)
    gm = fa.Model(cfg, w, dtype=dtype)
    om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=(dtype == "bf16"))
    T = 4300
    ids = synth.prompt_ids(cfg, T + 2, seed=13)
    gc, oc = gm.new_cache(T + 8), om.new_cache(T + 8)
    from test_gpu_parity import check_logits
    check_logits(gm.forward(gc, ids[:T], 0), om.forward(oc, ids[:T], 0), dtype, "4300-token prefill, window 4096")
    for i in range(T, T + 2):
        check_logits(gm.forward(gc, ids[i:i + 1], i), om.forward(oc, ids[i:i + 1], i), dtype, "decode after it")
    # and the same prompt cut by the library into chunks
    os.environ["FL_PREFILL_CHUNK"] = "1500"
    try:
        g2 = gm.new_cache(T + 8)
        o2 = om.new_cache(T + 8)
        check_logits(gm.forward(g2, ids[:T], 0), om.forward(o2, ids[:T], 0), dtype, "chunked 4300-token prefill, window 4096")
    finally:
        del os.environ["FL_PREFILL_CHUNK"]


def test_mistral_7b_long_prompt_chunked(env, monkeypatch):
    """A 20000-token prompt at full size (max_position_embeddings 32768, window 4096): the library cuts it into
    8192-token chunks; a different chunking gives the same logits, and decoding on from S = 20000 works."""
    torch, fa, bench = env
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = MODEL_CONFIGS["mistral-7b"]
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=12)
    gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
    del wts
    torch.cuda.empty_cache()
    T = 20000
    p = synth.prompt_ids(cfg, T, seed=6)
    c1 = gm.new_cache(T + 16)
    a = gm.forward(c1, p, 0)
    d1 = gm.forward(c1, [5], T)
    monkeypatch.setenv("FL_PREFILL_CHUNK", "3000")
    c2 = gm.new_cache(T + 16)
    b = gm.forward(c2, p, 0)
    close_bf16(b, a, "20000-token prompt, 3000- vs 8192-token chunks")
    close_bf16(gm.forward(c2, [5], T), d1, "decode at S = 20001")
    toks = gm.decode_greedy(c1, int(np.argmax(d1)), T + 1, 8)
    assert len(toks) == 8 and len(c1) == T + 9
    gm.close()


@pytest.mark.parametrize("name,T", [("qwen2-7b", 4096), ("mistral-7b", 4100),
                                    # mid-size prompts: the 128 x 256 kernel whose K slices meet inside the launch (k_gemm_h4.hip) runs the same epilogue
                                    ("mistral-7b", 512), ("mistral-7b", 300), ("qwen2-7b", 640)])
def test_long_prompt_residual_epilogue_equals_rmsnorm_launches(env, monkeypatch, name, T):
    """Long prompts: the 256x256 GEMM's residual epilogue (h += y, next norm's x*w, partial sums of squares -> 1/rms;
    EPI_RESID) against the separate rmsnorm_add launches it replaces (FL_GEMM_RESID=0), full width, 3 layers: the same
    math except the order in which a row's squares are summed.  T = 4100: a ragged last row tile, and o_proj / down_proj are
    peeled (whole rounds + a stream-K tail whose fix-up launch runs the same epilogue)."""
    torch, fa, bench = env
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS[name], num_hidden_layers=3)
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=11)
    gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
    del wts
    torch.cuda.empty_cache()
    ids = synth.prompt_ids(cfg, T, seed=17)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("FL_GEMM_RESID", mode)
        c = gm.new_cache(T + 8)
        gm.profile_begin()
        lg = gm.forward(c, ids, 0)
        names = {s["name"]: s["launches"] for s in gm.profile_end()}
        out[mode] = (lg, gm.forward(c, ids[:1], T), names)          # + one decode step on the cache the prefill left
        c.close()
    assert not any("finalize" in n or "resid" in n for n in out["0"][2]), out["0"][2]
    assert sum(v for n, v in out["1"][2].items() if "resid" in n) >= 2 * 3, out["1"][2]      # o_proj and down_proj of every layer
    if T <= 640:
        assert sum(v for n, v in out["1"][2].items() if "h4," in n and "resid" in n) == 2 * 3, out["1"][2]
    for k in (0, 1):
        a, b = out["1"][k], out["0"][k]
        # (noise floor of the bf16 pipeline: 3.5e-3 measured here; 5.6e-3 between the 256x256 and the 128x128 GEMM kernels, which
        # differ only in the order of the fp32 accumulation)
        assert np.linalg.norm(a - b) <= 1e-2 * np.linalg.norm(b), "rel L2 %.2e" % (np.linalg.norm(a - b) / np.linalg.norm(b))
        assert oracle.argmax(a) == oracle.argmax(b)
    gm.close()


_ORACLE_LONG = {}


@pytest.mark.parametrize("name,T,tp", [("mistral-7b", 1100, 1), ("qwen2-7b", 1100, 1), ("mistral-7b", 4100, 1), ("qwen2-7b", 4096, 1),
                                       ("mistral-7b", 512, 8), ("qwen2-7b", 4096, 4),       # these two: BASELINE configs C4 / C5, emulated ranks
                                       ("mistral-7b", 513, 1), ("mistral-7b", 545, 1)])     # one / 33 tokens past two row tiles: gate/up as 512 rows + a launch for the rest (k_linear.hip)
def test_7b_width_long_prompt_vs_oracle(env, name, T, tp):
    """The long-prompt kernels against the fp32 ORACLE (not against each other): full width, 2 layers (1 at 4096 / 4100 tokens, where
    the CPU side is 10 s per layer; the layer-to-layer hand-off at that length is held by the GPU-vs-GPU tests of this file, whose
    other side this test holds to the oracle at 1100 tokens).  1100 tokens (ragged):
    256x256 GEMMs in K slices (Qwen2: with the q/k/v bias riding in the RoPE launch), the key-split 32-row attention.  4096 /
    4100 tokens: peeled GEMMs with stream-K tails and the fix-up launch, the residual epilogue + rms_finalize, the snake
    schedule of the attention (4100: a ragged last block, Mistral's 4096-token window crossed).  Then four decode steps on
    the cache the prefill left."""
    import time
    torch, fa, bench = env
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS[name], num_hidden_layers=1 if T >= 4096 else 2)
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=5)
    w = host_copy(torch, wts)
    del wts
    torch.cuda.empty_cache()
    kw = {} if tp == 1 else dict(tp_mode=fa.binding.TP_EMULATED, tp_size=tp)      # every rank's shard on this GPU, real per-rank shapes
    g16 = fa.Model(cfg, w, dtype="bf16", **kw)
    o32 = oracle.OracleModel(cfg, w)
    ids = synth.prompt_ids(cfg, T + 4, seed=9)
    gc, oc = g16.new_cache(T + 16), o32.new_cache(T + 16)
    t0 = time.time()
    if (name, T) not in _ORACLE_LONG:                     # (the same seeded weights and ids for every tp: one oracle run per (model, length))
        _ORACLE_LONG[(name, T)] = [o32.forward(oc, ids[:T], 0)] + [o32.forward(oc, ids[i:i + 1], i) for i in range(T, T + 4)]
    refs = _ORACLE_LONG[(name, T)]
    ref = refs[0]
    print("oracle prefill of %d tokens: %.1f s" % (T, time.time() - t0))
    got = g16.forward(gc, ids[:T], 0)
    n = np.linalg.norm(ref)
    print("%s tp=%d prefill(%d) bf16 vs fp32 oracle: rel L2 %.2e, argmax %s" % (name, tp, T, np.linalg.norm(got - ref) / n, oracle.argmax(got) == oracle.argmax(ref)))
    # (measured 0.9-1.2e-2: bf16 storage against fp32 at full width; the same 2e-2 bound as bench.py's parity gate)
    assert np.linalg.norm(got - ref) <= 2e-2 * n, "%s prefill(%d): rel L2 %.4f" % (name, T, np.linalg.norm(got - ref) / n)
    assert oracle.argmax(got) == oracle.argmax(ref)
    for i in range(T, T + 4):
        ref, got = refs[1 + i - T], g16.forward(gc, ids[i:i + 1], i)
        n = np.linalg.norm(ref)
        assert np.linalg.norm(got - ref) <= 2e-2 * n, "%s decode at %d: rel L2 %.4f" % (name, i, np.linalg.norm(got - ref) / n)
    g16.close()


@pytest.mark.parametrize("name", ["mistral-7b", "qwen2-7b"])
def test_7b_width_long_prompt_vs_candle_emulation(env, name):
    """The production dtype at full width and a long prompt (1100 tokens, 2 layers): the product's bf16 logits are at least as
    close to fp32 as the oracle's candle-faithful bf16 emulation (round_bf16 = 2: rounding after every op, bf16 RoPE tables and
    positions for Mistral / Qwen) -- i.e. as the reference's own bf16 run would be -- and within that run's
    noise of it.  (tests/test_gpu_parity_bf16.py makes the same statement on the small HF-pinned models.)  Measured: the product
    is 0.9-1.2e-2 from fp32, the emulation ~1.0 (!): Mistral / Qwen build their RoPE angles from bf16 POSITIONS, which above 256
    are multiples of 2, 4, 8 ... (App. A.4) -- at 1100 tokens the high-frequency pairs of the reference's bf16 run are rotated
    by the wrong angle.  The product keeps fp32 tables; this test records the gap and bounds the product by it."""
    torch, fa, bench = env
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS[name], num_hidden_layers=2)
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=5)
    w = host_copy(torch, wts)
    del wts
    torch.cuda.empty_cache()
    T = 1100
    ids = synth.prompt_ids(cfg, T, seed=9)
    gm = fa.Model(cfg, w, dtype="bf16")
    gpu = gm.forward(gm.new_cache(T + 8), ids, 0)
    gm.close()
    o32, ocand = oracle.OracleModel(cfg, w), oracle.OracleModel(cfg, w, round_bf16=2)
    # (the fp32 oracle run is the one test_7b_width_long_prompt_vs_oracle[...-1100-1] made: same seeded weights, same first 1100 ids)
    ref = _ORACLE_LONG[(name, T)][0] if (name, T) in _ORACLE_LONG else o32.forward(o32.new_cache(T + 8), ids, 0)
    cand = ocand.forward(ocand.new_cache(T + 8), ids, 0)
    n = np.linalg.norm(ref)
    e_gpu, e_cand, d = np.linalg.norm(gpu - ref) / n, np.linalg.norm(cand - ref) / n, np.linalg.norm(gpu - cand) / n
    print("\n%s T=%d: rel L2 to fp32 -- gpu bf16 %.2e, candle-emulated bf16 %.2e; gpu vs candle-emulated %.2e" % (name, T, e_gpu, e_cand, d))
    assert e_gpu <= e_cand + 1e-3, (e_gpu, e_cand)
    assert d <= 1.5 * e_cand + 1e-3, (d, e_cand)


@pytest.mark.parametrize("name,T", [("mistral-7b", 512), ("mistral-7b", 300), ("qwen2-7b", 640), ("qwen2-7b", 384),
                                    ("mistral-7b", 200), ("qwen2-7b", 256)])     # 129-256 tokens: gate/up (no slices), QKV and o_proj on the kernel, down_proj stays
def test_mid_prompt_in_launch_slices_equal_the_slab_path(env, name, T):
    """257-640 tokens: the QKV projection with RoPE / bias / KV append in its epilogue and o_proj / down_proj with the residual
    epilogue, all on the 128 x 256 kernel whose K slices meet inside the launch (k_gemm_h4.hip), against the path it replaces
    (fp32 slabs summed by rope_kv / rmsnorm_add; gemm_h4 = 0): full width, 3 layers, the prefill's logits and a decode step on the
    cache it left (so the K / V the epilogue appended are read back).  Same math up to the order of the fp32 sums."""
    torch, fa, bench = env
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS[name], num_hidden_layers=3)
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=13)
    gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
    del wts
    torch.cuda.empty_cache()
    ids = synth.prompt_ids(cfg, T, seed=19)
    out = {}
    try:
        for mode in (1, 0, 2, 3):                       # 2: the in-launch path with the rms_finalize launches kept (rs_lazy = 0)
            fa.tune("gemm_h4", min(mode, 1)); fa.tune("rs_lazy", 0 if mode == 2 else 1)
            # 3: a plan that is wrong on purpose -- every projection is CLAIMED to take its row scales as partial sums, so those whose
            # kernel reads a vector only (Qwen2-7B's peeled gate/up with its stream-K tail) meet Launcher::rsp and must finish the sums
            # into the vector themselves (rs_parts_to_vector) instead of failing the forward: the launches and bits of mode 1
            fa.tune("debug_rs_parts", 1 if mode == 3 else 0)
            c = gm.new_cache(T + 8)
            gm.profile_begin()
            lg = gm.forward(c, ids, 0)
            names = {s["name"]: s["launches"] for s in gm.profile_end()}
            out[mode] = (lg, gm.forward(c, ids[:1], T), names)
            c.close()
    finally:
        fa.tune("reload_env", 0)
    nres = sum(v for n, v in out[1][2].items() if "h4," in n and "resid" in n)
    for k in (0, 1):        # (not the same bits: a kernel that CAN sum the partials but was not planned to now does; a stale vector would be O(1))
        assert np.linalg.norm(out[3][k] - out[1][k]) <= 1e-2 * np.linalg.norm(out[1][k]), "wrong plan: rel L2 %.2e" % (
            np.linalg.norm(out[3][k] - out[1][k]) / np.linalg.norm(out[1][k]))
    # 1/rms taken from the partial sums by the consuming projection (Launcher::rsp) against the rms_finalize launch: the same numbers
    # up to the order a row's partial sums are added in -- a kernel that ignored the request would read a stale vector
    # (what stays: the last layer's -- the final norm wants the vector -- and those in front of a projection whose kernel takes a
    # vector only, e.g. Qwen2-7B's peeled gate/up with its stream-K tail)
    assert sum(v for n, v in out[1][2].items() if "finalize" in n) < max(nres, 1), out[1][2]
    assert sum(v for n, v in out[2][2].items() if "finalize" in n) == nres, out[2][2]
    for k in (0, 1):
        assert np.linalg.norm(out[1][k] - out[2][k]) <= 1e-2 * np.linalg.norm(out[2][k]), "lazy row scales: rel L2 %.2e" % (    # (3.6e-3 measured: a last-bit 1/rms moves bf16 roundings downstream; a stale vector is O(1))
            np.linalg.norm(out[1][k] - out[2][k]) / np.linalg.norm(out[2][k]))
    assert not any("h4," in n for n in out[0][2]), out[0][2]
    assert sum(v for n, v in out[1][2].items() if "h4," in n and "rope" in n) == 3, out[1][2]
    assert (nres == 2 * 3) if T > 256 else (nres in (0, 3)), out[1][2]      # (<= 256 tokens: o_proj where its grid covers half the chip, never down_proj)
    if T <= 256 and name == "mistral-7b":               # (Qwen2-7B's gate/up is 296 of these tiles: more than one round, it stays on 256 x 256)
        assert sum(v for n, v in out[1][2].items() if n.startswith("gemm_mfma[h4,") and "sliced" not in n) == 3, out[1][2]     # gate/up
    assert not any("rope_kv" in n for n in out[1][2]), out[1][2]
    for k in (0, 1):
        a, b = out[1][k], out[0][k]
        assert np.linalg.norm(a - b) <= 1e-2 * np.linalg.norm(b), "rel L2 %.2e" % (np.linalg.norm(a - b) / np.linalg.norm(b))
        assert oracle.argmax(a) == oracle.argmax(b)
    gm.close()


@pytest.mark.parametrize("T", [512, 300, 2048, 600])
def test_gate_up_on_224_column_tiles_equals_the_256_column_grid(env, T):
    """Mistral-7B's fused gate/up matrix (28672 rows = 128 tiles of 224 = 112 of 256) on the 256 x 224 kernel (k_gemm_w14.hip) where
    that grid fills the chip: against the 256 x 256 grid, full width, 2 layers, prefill logits + a decode step.  Same K order, same
    epilogue arithmetic, same row scales from the partial sums (Launcher::rsp): the same numbers, up to the K order of the tiles the
    256-column grid splits between workgroups at 2048 tokens (stream-K: 3.5 rounds of the chip)."""
    torch, fa, bench = env
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS["mistral-7b"], num_hidden_layers=2)
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=31)
    gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
    del wts
    torch.cuda.empty_cache()
    ids = synth.prompt_ids(cfg, T, seed=37)
    out = {}
    try:
        for mode in (1, 0):
            fa.tune("gemm_w14", mode)
            c = gm.new_cache(T + 8)
            gm.profile_begin()
            lg = gm.forward(c, ids, 0)
            names = {s["name"]: s["launches"] for s in gm.profile_end()}
            out[mode] = (lg, gm.forward(c, ids[:1], T), names)
            c.close()
    finally:
        fa.tune("reload_env", 0)
    assert sum(v for n, v in out[1][2].items() if "w14," in n) == 2, out[1][2]
    assert not any("w14," in n for n in out[0][2]), out[0][2]
    for k in (0, 1):
        a, b = out[1][k], out[0][k]
        assert np.linalg.norm(a - b) <= 1e-2 * np.linalg.norm(b), "rel L2 %.2e" % (np.linalg.norm(a - b) / np.linalg.norm(b))
        assert oracle.argmax(a) == oracle.argmax(b)
    gm.close()


@pytest.mark.parametrize("name,T", [("mistral-7b", 2048), ("mistral-7b", 1900), ("qwen2-7b", 4096), ("mistral-7b", 4096)])
def test_long_prompt_rope_in_the_qkv_epilogue_equals_the_rope_launch(env, name, T):
    """Long prompts: RoPE, the q/k/v bias and the KV append in the epilogue of the four-wave 256 x 256 kernel (Qwen2-7B at 4096 tokens:
    whole rounds there + the 512 tail columns -- its value heads -- on the 128 x 256 kernel with in-launch slices) against the fp32 QKV
    matrix + rope_kv_append launch they replace: full width, 2 layers, prefill logits and a decode step on the cache the epilogue wrote."""
    torch, fa, bench = env
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS[name], num_hidden_layers=2)
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=41)
    gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
    del wts
    torch.cuda.empty_cache()
    ids = synth.prompt_ids(cfg, T, seed=43)
    out = {}
    try:
        for mode in (1, 0):
            fa.tune("gemm_rope_4w", mode)
            c = gm.new_cache(T + 8)
            gm.profile_begin()
            lg = gm.forward(c, ids, 0)
            names = {s["name"]: s["launches"] for s in gm.profile_end()}
            out[mode] = (lg, gm.forward(c, ids[:1], T), names)
            c.close()
    finally:
        fa.tune("reload_env", 0)
    assert sum(v for n, v in out[1][2].items() if "4w," in n and "rope" in n) == 2, out[1][2]
    assert not any("rope_kv" in n for n in out[1][2]), out[1][2]
    assert sum(v for n, v in out[0][2].items() if "rope_kv" in n) == 2, out[0][2]
    if T == 4096:                                        # the peeled tail columns: Qwen2-7B's 512 (4 slices), Mistral-7B's 2048 (unsliced, 256 tiles)
        assert sum(v for n, v in out[1][2].items() if "h4," in n and "rope" in n) == 2, out[1][2]
    for k in (0, 1):
        a, b = out[1][k], out[0][k]
        assert np.linalg.norm(a - b) <= 1e-2 * np.linalg.norm(b), "rel L2 %.2e" % (np.linalg.norm(a - b) / np.linalg.norm(b))
        assert oracle.argmax(a) == oracle.argmax(b)
    gm.close()


def test_selection_thresholds_from_both_sides(env):
    """The prefill's kernel-selection thresholds (attention forms at 352 / 576 / 640, the 128 x 256 kernel's 256 / 640, the 224-column
    grid's 512 / 768, the non-temporal rule's 176, the RoPE epilogue's 768, ...) from both sides: the default selection against the
    conservative one (round-4 kernels and rules off) on the same prompt, Mistral-7B width, 2 layers, prefill logits + a decode step.
    tools/policy_boundaries.py is the long form (three models, 39 lengths; profiles/r04/policy_boundaries.txt)."""
    torch, fa, bench = env
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS["mistral-7b"], num_hidden_layers=2)
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=3)
    gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
    del wts
    torch.cuda.empty_cache()
    off = {"gemm_h4": 0, "gemm_w14": 0, "gemm_rope_4w": 0, "attn_pf32_ks2": 1, "attn_pf32_min_t": 640, "h4_nt": 0, "w14_nt": 0, "skinny_nt": 0, "rs_lazy": 0}
    rs = np.random.RandomState(7)
    try:
        for T in (175, 176, 256, 257, 351, 352, 512, 513, 576, 577, 639, 640, 641, 767, 768, 1025):
            ids = rs.randint(0, cfg["vocab_size"], size=T + 1).astype(np.uint32)
            out = []
            for conservative in (False, True):
                fa.tune("reload_env", 0)
                if conservative:
                    for k, v in off.items():
                        fa.tune(k, v)
                c = gm.new_cache(T + 8)
                out.append((gm.forward(c, ids[:T], 0), gm.forward(c, ids[T:T + 1], T)))
                c.close()
            for k in (0, 1):
                a, b = out[0][k], out[1][k]
                assert np.linalg.norm(a - b) <= 1.5e-2 * np.linalg.norm(b), "T=%d %s: rel L2 %.2e" % (T, ("prefill", "decode")[k], np.linalg.norm(a - b) / np.linalg.norm(b))
    finally:
        fa.tune("reload_env", 0)
    gm.close()


@pytest.mark.parametrize("T", [512, 640])
def test_mid_prompt_small_hidden_size_on_the_in_launch_kernel(env, T):
    """TinyLlama-1.1B's widths (h = 2048, d = 64, I = 5632) at 257-640 tokens: gate/up without slices, QKV + RoPE (two 64-wide heads
    per wave column: the d = 64 form of the RoPE epilogue), o_proj and down_proj with the residual epilogue -- against the path they
    replace, full width, 3 layers, prefill logits and a decode step on the cache the epilogue appended to."""
    torch, fa, bench = env
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS["tinyllama-1.1b"], num_hidden_layers=3)
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=23)
    gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
    del wts
    torch.cuda.empty_cache()
    ids = synth.prompt_ids(cfg, T, seed=29)
    out = {}
    try:
        for mode in (1, 0):
            fa.tune("gemm_h4", mode)
            c = gm.new_cache(T + 8)
            gm.profile_begin()
            lg = gm.forward(c, ids, 0)
            names = {s["name"]: s["launches"] for s in gm.profile_end()}
            out[mode] = (lg, gm.forward(c, ids[:1], T), names)
            c.close()
    finally:
        fa.tune("reload_env", 0)
    assert sum(v for n, v in out[1][2].items() if "h4," in n and "rope" in n) == 3, out[1][2]
    assert sum(v for n, v in out[1][2].items() if "h4," in n and "resid" in n) >= 3, out[1][2]
    assert not any("h4," in n for n in out[0][2]), out[0][2]
    for k in (0, 1):
        a, b = out[1][k], out[0][k]
        assert np.linalg.norm(a - b) <= 1e-2 * np.linalg.norm(b), "rel L2 %.2e" % (np.linalg.norm(a - b) / np.linalg.norm(b))
        assert oracle.argmax(a) == oracle.argmax(b)
    gm.close()


@needs_experimental
@pytest.mark.parametrize("name,T", [("mistral-7b", 2), ("mistral-7b", 16), ("mistral-7b", 33), ("mistral-7b", 100), ("mistral-7b", 128),
                                    ("qwen2-7b", 24), ("qwen2-7b", 128), ("tinyllama-1.1b", 7), ("tinyllama-1.1b", 128)])
def test_short_prompt_fused_layer_equals_the_slab_path(env, name, T):
    """2-128 tokens: QKV with RoPE / bias / KV append in its epilogue, o_proj and down_proj with the residual + norm epilogue, gate/up
    with its row scales from the partial sums -- all on the weight-streaming kernel whose K slices meet inside the launch
    (k_gemm_skf.hip, FL_GEMM_SKF=2): five launches per layer.  Against the default path (FL_GEMM_SKF=0 / 1: fp32 slabs summed by rope_kv / rmsnorm_add,
    eight launches): full width, 3 layers, the prefill's logits and two decode steps on the cache the epilogue wrote.  Same math up to
    the order of fp32 sums (1/rms) and of `acc * rs` against the slabs' sum."""
    torch, fa, bench = env
    from fastllm_amd.configs import MODEL_CONFIGS
    cfg = dict(MODEL_CONFIGS[name], num_hidden_layers=3)
    wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0), seed=17)
    gm = fa.Model(cfg, bench.as_fl_tensors(wts, 0), dtype="bf16")
    del wts
    torch.cuda.empty_cache()
    ids = synth.prompt_ids(cfg, T + 2, seed=23)
    out = {}
    try:
        for mode in (1, 0):
            fa.tune("gemm_skf", 2 if mode else 0)            # (2: the opt-in five-launch layer; it measured slower than the slab path, profiles/r05)
            c = gm.new_cache(T + 8)
            gm.profile_begin()
            lg = gm.forward(c, ids[:T], 0)
            names = {s["name"]: s["launches"] for s in gm.profile_end()}
            out[mode] = (lg, gm.forward(c, ids[T:T + 1], T), gm.forward(c, ids[T + 1:T + 2], T + 1), names)
            c.close()
    finally:
        fa.tune("reload_env", 0)
    n1, n0 = out[1][3], out[0][3]
    assert sum(v for n, v in n1.items() if "skf," in n and "rope" in n) == 3, n1
    assert sum(v for n, v in n1.items() if "skf," in n and "resid" in n) == 6, n1
    assert sum(v for n, v in n1.items() if "skf," in n and "glu" in n) == 3, n1
    import json
    assert not any("rope_kv" in n for n in n1) and n1.get("rmsnorm_add", 0) == 1, json.dumps(n1)     # (the first layer's)
    assert sum(n1.values()) <= 5 * 3 + 6, json.dumps(n1)     # five launches per layer + embed, first norm, final 1/rms, lm_head, token selection
    assert not any("skf," in n for n in n0), n0
    for k in (0, 1, 2):
        a, b = out[1][k], out[0][k]
        assert np.linalg.norm(a - b) <= 1e-2 * np.linalg.norm(b), "call %d: rel L2 %.2e" % (k, np.linalg.norm(a - b) / np.linalg.norm(b))
        assert oracle.argmax(a) == oracle.argmax(b)
    gm.close()
