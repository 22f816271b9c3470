"""GPU tests of the C++ host mirror: Model<M>::generate semantics through the trait-shaped layer."""
import ctypes as C

import numpy as np
import pytest

import synth
from oracle import oracle
from test_host_mirror import config_json, host  # noqa: F401

pytestmark = pytest.mark.gpu
FAM = {"llama": 0, "mistral": 1, "qwen2": 2}


def make(host, name, dtype=0):
    from fastllm_amd import binding
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    arr = (binding.FlTensor * len(w))()
    keep = []
    for i, (k, a) in enumerate(w.items()):
        a = np.ascontiguousarray(a)
        keep.append(a)
        arr[i].name, arr[i].dtype, arr[i].ndim, arr[i].data, arr[i].device = k.encode(), 1, a.ndim, a.ctypes.data, -1
        for j, s in enumerate(a.shape):
            arr[i].shape[j] = s
    h = C.c_void_p()
    rc = host.flh_model_create(FAM[cfg["family"]], config_json(cfg), arr, len(w), dtype, 0, C.byref(h))
    assert rc == 0, host.flh_last_error()
    return h, cfg, w


def generate(host, h, prompt, n, eos=-1, temperature=0.0):
    prompt = np.ascontiguousarray(prompt, dtype=np.uint32)
    out = np.zeros(n, dtype=np.uint32)
    n_out, fw = C.c_size_t(0), C.c_size_t(0)
    rc = host.flh_generate(h, prompt.ctypes.data, prompt.size, n, temperature, eos, out.ctypes.data, C.byref(n_out), C.byref(fw))
    assert rc == 0, host.flh_last_error()
    return out[: n_out.value], fw.value


@pytest.mark.parametrize("name", ["llama_a", "mistral_a", "qwen2_a"])
@pytest.mark.parametrize("mode", ["reference", "tokens"])
def test_generate_matches_oracle_generate(host, name, mode, monkeypatch):
    """Model<M>::generate (mod.rs:363-463) in fp32: identical token ids; Mistral/Qwen in `reference` mode
    reproduce the call-counter RoPE offset (quirk C.1)."""
    monkeypatch.setenv("FASTLLM_POS_MODE", mode)
    monkeypatch.setenv("FASTLLM_MAX_SEQ", "64")
    h, cfg, w = make(host, name, dtype=0)
    om = oracle.OracleModel(cfg, synth.as_f32(w))
    prompt = synth.prompt_ids(cfg, 8)
    want = om.generate(om.new_cache(64), prompt, 12, pos_mode=mode)
    got, forwards = generate(host, h, prompt, 12)
    np.testing.assert_array_equal(got, want)
    assert forwards == 1 + 12                      # prefill + one forward per token, incl. the wasted last one (C.5)
    # a second request starts from a fresh cache (mod.rs:370)
    got2, _ = generate(host, h, prompt, 12)
    np.testing.assert_array_equal(got2, want)
    host.flh_model_destroy(h)


def test_generate_with_temperature_host_and_device_samplers_agree(host, monkeypatch):
    """temperature > 0 (mod.rs:373-374): the mirror's host-side LogitsProcessor over fl_forward logits -- the
    reference's own loop shape -- and the device-side sampler (fl_forward_sample + fl_decode_sample) draw the
    same tokens from the same seed-0 stream."""
    import fastllm_amd as fa
    monkeypatch.setenv("FASTLLM_POS_MODE", "tokens")
    monkeypatch.setenv("FASTLLM_MAX_SEQ", "64")
    h, cfg, w = make(host, "llama_a", dtype=0)
    prompt = synth.prompt_ids(cfg, 8)
    got, forwards = generate(host, h, prompt, 16, temperature=0.8)
    assert forwards == 17 and len(got) == 16
    greedy, _ = generate(host, h, prompt, 16)
    assert not np.array_equal(got, greedy)                      # it does sample
    again, _ = generate(host, h, prompt, 16, temperature=0.8)   # a fresh seed-0 processor per request
    np.testing.assert_array_equal(again, got)
    gm = fa.Model(cfg, w, dtype="f32")
    c = gm.new_cache(64)
    first = gm.forward_sample(c, prompt, 0, 0.8)
    rest = gm.decode_sample(c, first, len(prompt), 15, 0.8, draws_done=1)
    np.testing.assert_array_equal(np.concatenate([[first], rest]).astype(np.uint32), got)
    host.flh_model_destroy(h)


def test_eos_stops_before_emitting(host, monkeypatch):
    monkeypatch.setenv("FASTLLM_MAX_SEQ", "64")
    h, cfg, w = make(host, "llama_a", dtype=0)
    prompt = synth.prompt_ids(cfg, 8)
    full, _ = generate(host, h, prompt, 10)
    eos = int(full[4])
    first_hit = int(np.argmax(full == eos))
    got, forwards = generate(host, h, prompt, 10, eos=eos)
    np.testing.assert_array_equal(got, full[:first_hit])        # mod.rs:431-436: break before push
    assert forwards == 1 + first_hit
    host.flh_model_destroy(h)


def test_trait_forward_counter_semantics(host, monkeypatch):
    """MistralWithConfig::forward ignores pos, uses and advances the per-call counter (mistral.rs:206-236)."""
    monkeypatch.setenv("FASTLLM_POS_MODE", "reference")
    monkeypatch.setenv("FASTLLM_MAX_SEQ", "64")
    h, cfg, w = make(host, "mistral_a", dtype=0)
    om = oracle.OracleModel(cfg, synth.as_f32(w))
    oc = om.new_cache(64)
    ids = synth.prompt_ids(cfg, 7)
    lg = np.zeros(cfg["vocab_size"], np.float32)
    n = C.c_size_t(0)
    assert host.flh_cache_offset(h) == 0
    assert host.flh_forward(h, ids[:6].ctypes.data, 6, 12345, lg.ctypes.data, C.byref(n)) == 0    # pos is ignored
    assert n.value == cfg["vocab_size"] and host.flh_cache_offset(h) == 1                        # +1 per call, not +T
    np.testing.assert_allclose(lg, om.forward(oc, ids[:6], 0), atol=1e-3, rtol=0)
    assert host.flh_forward(h, ids[6:7].ctypes.data, 1, 999, lg.ctypes.data, C.byref(n)) == 0
    np.testing.assert_allclose(lg, om.forward(oc, ids[6:7], 1), atol=1e-3, rtol=0)               # rotated as position 1
    assert host.flh_cache_offset(h) == 2
    host.flh_cache_reset(h)
    assert host.flh_cache_offset(h) == 0
    host.flh_model_destroy(h)


@pytest.mark.parametrize("oneshot,graph", [("0", "0"), ("0", "1"), ("1", "0"), ("1", "1")])
def test_rccl_plumbing_single_rank(monkeypatch, oneshot, graph):
    """A 1-rank RCCL communicator at the real all-reduce call sites (identity) must not change results.
    oneshot=1 also runs the inbox bootstrap a multi-process group does over RCCL (handle all-gather, connect,
    self-test against ncclAllReduce, vote) for a group of one, and then the one-shot kernels at the call sites."""
    import fastllm_amd as fa
    monkeypatch.setenv("FL_ONESHOT", oneshot)
    monkeypatch.setenv("FL_GRAPH", graph)
    cfg = synth.CONFIGS["llama_a"]
    w = synth.synth_weights(cfg)
    ids = synth.prompt_ids(cfg, 10)
    plain = fa.Model(cfg, w, dtype="f32")
    c = plain.new_cache(32)
    a1 = plain.forward(c, ids[:9], 0)
    a2 = plain.forward(c, ids[9:], 9)
    monkeypatch.setenv("FL_DEBUG_RCCL_SELF", "1")
    m = fa.Model(cfg, w, dtype="f32")
    assert m.info().small_collectives == (2 if oneshot == "1" else 0)
    # ... and the exchange fused into the GEMV epilogues passed its self-test and vote (comm_ll.h); the decode steps below use it
    assert m.info().fused_all_reduce == (1 if oneshot == "1" else 0)
    c = m.new_cache(32)
    np.testing.assert_array_equal(m.forward(c, ids[:9], 0), a1)
    np.testing.assert_array_equal(m.forward(c, ids[9:], 9), a2)
    c0 = plain.new_cache(32)
    plain.forward(c0, ids[:9], 0); plain.forward(c0, ids[9:], 9)
    np.testing.assert_array_equal(m.decode_greedy(c, 3, 10, 5), plain.decode_greedy(c0, 3, 10, 5))


TOKEN_CB = C.CFUNCTYPE(C.c_int, C.c_uint32, C.c_void_p)


def generate_stream(host, h, prompt, n, eos=-1, temperature=0.0, stop_after=None):
    """flh_generate_stream: tokens as the callback receives them; stop_after = the receiver hangs up after that many"""
    prompt = np.ascontiguousarray(prompt, dtype=np.uint32)
    got = []

    def on_token(tok, _user):
        got.append(int(tok))
        return 0 if stop_after is not None and len(got) >= stop_after else 1
    cb = TOKEN_CB(on_token)
    fw = C.c_size_t(0)
    host.flh_generate_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.c_int64, TOKEN_CB, C.c_void_p, C.POINTER(C.c_size_t)]
    rc = host.flh_generate_stream(h, prompt.ctypes.data, prompt.size, n, temperature, eos, cb, None, C.byref(fw))
    assert rc == 0, host.flh_last_error()
    return np.array(got, dtype=np.uint32), fw.value


@pytest.mark.parametrize("name", ["llama_a", "mistral_a", "qwen2_a"])
def test_generate_stream_matches_generate_and_the_oracle(host, name, monkeypatch):
    """generate_stream / generate_tokens_inner (mod.rs:137-238, 268-340): the stream's tokens are generate()'s tokens, one
    callback per token before the next forward; a receiver that hangs up ends the loop before another forward; EOS ends it
    before the token is emitted."""
    monkeypatch.setenv("FASTLLM_POS_MODE", "reference")
    monkeypatch.setenv("FASTLLM_MAX_SEQ", "64")
    h, cfg, w = make(host, name, dtype=0)
    om = oracle.OracleModel(cfg, synth.as_f32(w))
    prompt = synth.prompt_ids(cfg, 8)
    want = om.generate(om.new_cache(64), prompt, 12, pos_mode="reference")
    got, fw = generate_stream(host, h, prompt, 12)
    np.testing.assert_array_equal(got, want)
    assert fw == 1 + 12
    np.testing.assert_array_equal(generate(host, h, prompt, 12)[0], want)          # the blocking path is untouched by a stream
    part, fw = generate_stream(host, h, prompt, 12, stop_after=5)
    np.testing.assert_array_equal(part, want[:5])
    assert fw == 1 + 4                                                             # no forward behind the token nobody received
    eos = int(want[3])
    cut, fw = generate_stream(host, h, prompt, 12, eos=eos)
    first = int(np.flatnonzero(want == eos)[0])
    np.testing.assert_array_equal(cut, want[:first])
    assert fw == 1 + first
    host.flh_model_destroy(h)


def test_concurrent_streams_share_one_model(host, monkeypatch):
    """The reference spawns one task per stream over a clone of the model (mod.rs:155-160): two streams from two threads on ONE
    handle, different prompts, each with its own cache -- the same tokens as one after the other."""
    import threading
    monkeypatch.setenv("FASTLLM_POS_MODE", "tokens")
    monkeypatch.setenv("FASTLLM_MAX_SEQ", "96")
    h, cfg, w = make(host, "mistral_a", dtype=1)
    prompts = [synth.prompt_ids(cfg, 8, seed=s) for s in (3, 4, 5, 6)]
    alone = [generate_stream(host, h, p, 40)[0] for p in prompts]
    res = [None] * len(prompts)

    def run(i):
        res[i] = generate_stream(host, h, prompts[i], 40)[0]
    ts = [threading.Thread(target=run, args=(i,)) for i in range(len(prompts))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for a, b in zip(alone, res):
        np.testing.assert_array_equal(a, b)
    host.flh_model_destroy(h)


BATCH_TOKEN_CB = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_uint32, C.c_void_p)
BATCH_DONE_CB = C.CFUNCTYPE(None, C.c_uint64, C.c_size_t, C.c_void_p)


def run_batcher(host, h, requests, slots, chunk=4, max_seq=96):
    """requests: (prompt, max_tokens, temperature, eos, stop_after) -> per request the tokens its callback received, + (steps, prefills)"""
    host.flh_batcher_create.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
    host.flh_batcher_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.c_int64, BATCH_TOKEN_CB, BATCH_DONE_CB, C.c_void_p,
                                        C.POINTER(C.c_uint64)]
    host.flh_batcher_run.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    host.flh_batcher_destroy.argtypes = [C.c_void_p]
    b = C.c_void_p()
    assert host.flh_batcher_create(h, slots, max_seq, chunk, C.byref(b)) == 0, host.flh_last_error()
    got, done, stop = {}, {}, {}

    def on_token(rid, tok, _user):
        got[rid].append(int(tok))
        return 0 if stop[rid] is not None and len(got[rid]) >= stop[rid] else 1

    def on_done(rid, n, _user):
        done[rid] = int(n)
    cb, dcb = BATCH_TOKEN_CB(on_token), BATCH_DONE_CB(on_done)
    ids = []
    for prompt, n, temp, eos, stop_after in requests:
        p = np.ascontiguousarray(prompt, dtype=np.uint32)
        rid = C.c_uint64(0)
        # (the id is known before any callback runs: requests are admitted inside flh_batcher_run)
        assert host.flh_batcher_submit(b, p.ctypes.data, p.size, n, temp, eos, cb, dcb, None, C.byref(rid)) == 0, host.flh_last_error()
        got[rid.value], stop[rid.value] = [], stop_after
        ids.append(rid.value)
    steps, pre = C.c_size_t(0), C.c_size_t(0)
    assert host.flh_batcher_run(b, C.byref(steps), C.byref(pre)) == 0, host.flh_last_error()
    host.flh_batcher_destroy(b)
    for rid in ids:
        assert done.get(rid) == len(got[rid]), "request %d: done callback %s, tokens %d" % (rid, done.get(rid), len(got[rid]))
    return [np.array(got[r], dtype=np.uint32) for r in ids], steps.value, pre.value


@pytest.mark.parametrize("name,dtype,mode", [("llama_a", 0, "reference"), ("mistral_a", 0, "reference"), ("qwen2_a", 0, "tokens"), ("mistral_a", 1, "tokens")])
def test_stream_batcher_serves_the_streams_of_generate_stream(host, name, dtype, mode, monkeypatch):
    """StreamBatcher (continuous batching over mod.rs:137-238's concurrent streams): nine requests -- different prompts,
    lengths, max_tokens, an EOS id, a receiver that hangs up, two with a temperature -- through THREE slots, four steps per chunk.
    Every request's tokens are what flh_generate_stream gives it alone (greedy: exactly; the sampled ones in fp32, where the batch's
    logits are the single stream's to 1e-6), far fewer batch steps than the streams' forwards add up to, one prefill per request."""
    monkeypatch.setenv("FASTLLM_POS_MODE", mode)
    monkeypatch.setenv("FASTLLM_MAX_SEQ", "96")
    h, cfg, w = make(host, name, dtype=dtype)
    rs = np.random.RandomState(5)
    reqs = []
    for i in range(9):
        prompt = synth.prompt_ids(cfg, 3 + int(rs.randint(0, 12)), seed=40 + i)
        reqs.append([prompt, 5 + int(rs.randint(0, 30)), 0.0, -1, None])
    reqs[2][4] = 3                                            # the receiver hangs up after three tokens
    alone = [generate_stream(host, h, r[0], r[1], eos=r[3], temperature=r[2], stop_after=r[4])[0] for r in reqs]
    reqs[4][3] = int(alone[4][min(4, len(alone[4]) - 1)])      # an EOS id that comes up a few tokens in
    alone[4] = generate_stream(host, h, reqs[4][0], reqs[4][1], eos=reqs[4][3])[0]
    if dtype == 0:
        for i in (6, 7):
            reqs[i][2] = 0.8
            alone[i] = generate_stream(host, h, reqs[i][0], reqs[i][1], temperature=0.8)[0]
    got, steps, prefills = run_batcher(host, h, [tuple(r) for r in reqs], slots=3)
    for i in range(9):
        np.testing.assert_array_equal(got[i], alone[i], err_msg="request %d" % i)
    assert prefills == 9
    total = sum(len(a) for a in alone)
    assert steps < total                                      # the streams shared their steps (three at a time)
    assert steps >= total // 3
    # a one-slot batcher is the streams one after the other
    got1, _, _ = run_batcher(host, h, [tuple(r) for r in reqs[:3]], slots=1)
    for i in range(3):
        np.testing.assert_array_equal(got1[i], alone[i])
    host.flh_model_destroy(h)
