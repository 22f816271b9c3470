"""Multi-process tensor parallelism with the one-shot collectives (fastllm_amd/csrc/k_comm.hip).

tp ranks are separate processes that here share the box's single GPU: the inboxes are real
hipIpc mappings between processes and the flag / epoch protocol runs exactly as it does across
xGMI; only the link is missing.  Every rank must produce bit-identical logits (the sum runs in rank
order everywhere), equal to FL_TP_EMULATED's (same shards, same order) and within tolerance of the
oracle.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import synth
from oracle import oracle
from test_gpu_parity import check_logits

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def run_group(tmp_path, name, dtype, tp, T=10, n_tf=4, n_greedy=6, env_extra=None, ranks_per_proc=None):
    """tp ranks as tp / ranks_per_proc worker processes.  The GPU boxes admit at most six processes on the card, so groups of more
    than four ranks run two ranks (threads) per process: every rank still reaches six of its seven peers through hipIpc mappings."""
    rpp = ranks_per_proc or (1 if tp <= 4 else 2)
    assert tp % rpp == 0 and tp // rpp <= 5
    env = dict(os.environ, FL_AR_TIMEOUT_MS="8000")
    env.update(env_extra or {})
    groups = [",".join(str(r) for r in range(p * rpp, (p + 1) * rpp)) for p in range(tp // rpp)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "tp_ipc_worker.py"), g, str(tp), name, dtype,
                               str(tmp_path), str(T), str(n_tf), str(n_greedy)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for g in groups]
    outs = []
    try:
        for p in procs:
            o, _ = p.communicate(timeout=240)
            outs.append(o.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for g, p, o in zip(groups, procs, outs):
        assert p.returncode == 0, "ranks %s failed:\n%s" % (g, o[-3000:])
    return [np.load(os.path.join(str(tmp_path), "out_%d.npz" % r)) for r in range(tp)]


@pytest.mark.parametrize("name,tp,dtype", [("llama_a", 2, "bf16"), ("mistral_a", 2, "f32"), ("qwen2_a", 2, "bf16"),
                                           ("llama_mha", 2, "f32"), ("llama_tp4", 4, "bf16"), ("llama_tp8", 8, "bf16"),
                                           ("llama_tp8", 8, "f32"), ("llama_tp4", 4, "f32")])
def test_multiprocess_oneshot_matches_emulated_and_oracle(tmp_path, name, tp, dtype):
    import fastllm_amd as fa
    from fastllm_amd import binding
    cfg = synth.CONFIGS[name]
    T, n_tf, n_greedy = 10, 4, 6
    res = run_group(tmp_path, name, dtype, tp, T, n_tf, n_greedy)
    for r in range(1, tp):                                   # lock step: every rank holds the same bits
        for k in ("prefill", "decode", "tokens"):
            np.testing.assert_array_equal(res[r][k], res[0][k], err_msg="rank %d vs 0: %s" % (r, k))
    w = synth.synth_weights(cfg)
    ids = synth.prompt_ids(cfg, T + n_tf, seed=11)
    gN = fa.Model(cfg, w, dtype=dtype, tp_mode=binding.TP_EMULATED, tp_size=tp)
    om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=(dtype == "bf16"))
    cN, oc = gN.new_cache(64), om.new_cache(64)
    e, o = gN.forward(cN, ids[:T], 0), om.forward(oc, ids[:T], 0)
    np.testing.assert_array_equal(res[0]["prefill"], e)
    check_logits(res[0]["prefill"], o, dtype, "multi-process prefill vs oracle")
    for i in range(n_tf):
        e, o = gN.forward(cN, ids[T + i:T + i + 1], T + i), om.forward(oc, ids[T + i:T + i + 1], T + i)
        np.testing.assert_array_equal(res[0]["decode"][i], e)
        check_logits(res[0]["decode"][i], o, dtype, "multi-process decode vs oracle")
    f = gN.forward_argmax(cN, ids[:1], T + n_tf)
    rest = gN.decode_greedy(cN, f, T + n_tf + 1, n_greedy)
    np.testing.assert_array_equal(res[0]["tokens"], np.concatenate([[f], rest]).astype(np.uint32))
    cN.close()
    gN.close()


def test_small_inbox_chunks_collectives(tmp_path):
    """An inbox smaller than the message: prefill all-reduces and the logits all-gather run as several
    one-shot rounds."""
    res = run_group(tmp_path, "mistral_a", "bf16", 2, env_extra={"FL_AR_INBOX_FLOATS": "64"})
    ref_dir = tmp_path / "ref"
    ref_dir.mkdir()
    ref = run_group(ref_dir, "mistral_a", "bf16", 2)
    for k in ("prefill", "decode", "tokens"):
        np.testing.assert_array_equal(res[0][k], ref[0][k])
        np.testing.assert_array_equal(res[1][k], ref[0][k])


def test_unconnected_group_fails_loudly():
    import fastllm_amd as fa
    from fastllm_amd import binding
    cfg = synth.CONFIGS["llama_a"]
    m = fa.Model(cfg, synth.synth_weights(cfg), tp_mode=binding.TP_MULTI_PROCESS, tp_size=2, tp_rank=0, device_ids=[0])
    c = m.new_cache(16)
    with pytest.raises(fa.FastLLMError) as e:
        m.forward(c, [1, 2, 3], 0)
    msg = str(e.value)
    del e                                    # (the exception info holds the frame, i.e. the model: it would live until some later gc pass)
    assert "not connected" in msg
    c.close()
    m.close()


@pytest.mark.parametrize("name,tp,dtype", [("llama_a", 2, "bf16"), ("qwen2_a", 2, "f32"), ("mistral_a", 2, "bf16")])
def test_single_process_group_on_one_device(name, tp, dtype):
    """FL_TP_SINGLE_PROCESS (one process drives all shards: the reference's process model) with every shard on
    device 0: the one-shot collectives over plain peer pointers, one hipGraph per shard replayed side by side.
    Same partition and summation order as FL_TP_EMULATED, so the results must be the same bits.  (Two shards only:
    on ONE device the shards' streams share the process's few hardware queues, and a shard whose launches queue
    behind another shard's waiting collective can never signal it; with one device per shard that cannot happen.)"""
    import gc
    import fastllm_amd as fa
    from fastllm_amd import binding
    # models that earlier tests left to the garbage collector still own streams: on ONE device this process's streams share a few
    # hardware queues, and the two shards' streams must not land on one (tools/sp_probe.py: a third model alive -> a shard waits
    # for a peer whose launches sit behind it in the same queue)
    gc.collect()
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    gS = fa.Model(cfg, w, dtype=dtype, tp_mode=binding.TP_SINGLE_PROCESS, tp_size=tp, device_ids=[0] * tp)
    gE = fa.Model(cfg, w, dtype=dtype, tp_mode=binding.TP_EMULATED, tp_size=tp)
    assert gS.info().small_collectives == 2
    ids = synth.prompt_ids(cfg, 14, seed=11)
    cS, cE = gS.new_cache(64), gE.new_cache(64)
    np.testing.assert_array_equal(gS.forward(cS, ids[:10], 0), gE.forward(cE, ids[:10], 0))
    for i in range(10, 14):
        np.testing.assert_array_equal(gS.forward(cS, ids[i:i + 1], i), gE.forward(cE, ids[i:i + 1], i))
    f = gS.forward_argmax(cS, ids[:1], 14)
    assert f == gE.forward_argmax(cE, ids[:1], 14)
    np.testing.assert_array_equal(gS.decode_greedy(cS, f, 15, 20), gE.decode_greedy(cE, f, 15, 20))
    assert len(cS) == 35
    gS.close()
    gE.close()


def test_overlapped_prefill_multiprocess_and_single_process(tmp_path, monkeypatch):
    """Prompts of FL_TP_OVERLAP_MIN_T tokens and more take the two-chunk prefill whose all-reduces run on a side
    stream while the other chunk computes: same rows, same sums -> the same bits as the plain schedule
    (FL_TP_EMULATED), in a 2-rank multi-process group and in a 2-shard single-process group."""
    import fastllm_amd as fa
    from fastllm_amd import binding
    name, tp, dtype, T = "mistral_a", 2, "bf16", 400
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    ids = synth.prompt_ids(cfg, T + 4, seed=11)
    gE = fa.Model(cfg, w, dtype=dtype, tp_mode=binding.TP_EMULATED, tp_size=tp)
    cE = gE.new_cache(512)
    want = gE.forward(cE, ids[:T], 0)
    want_d = gE.forward(cE, ids[T:T + 1], T)
    res = run_group(tmp_path, name, dtype, tp, T=T, n_tf=4, n_greedy=4, env_extra={"FL_TP_OVERLAP_MIN_T": "256"})
    for r in range(tp):
        np.testing.assert_array_equal(res[r]["prefill"], want)
        np.testing.assert_array_equal(res[r]["decode"][0], want_d)
    monkeypatch.setenv("FL_TP_OVERLAP_MIN_T", "256")
    gS = fa.Model(cfg, w, dtype=dtype, tp_mode=binding.TP_SINGLE_PROCESS, tp_size=tp, device_ids=[0] * tp)
    cS = gS.new_cache(512)
    np.testing.assert_array_equal(gS.forward(cS, ids[:T], 0), want)
    np.testing.assert_array_equal(gS.forward(cS, ids[T:T + 1], T), want_d)
    # and against the same group with the overlap switched off
    monkeypatch.setenv("FL_TP_OVERLAP_MIN_T", "100000")
    cS2 = gS.new_cache(512)
    np.testing.assert_array_equal(gS.forward(cS2, ids[:T], 0), want)


@pytest.mark.parametrize("name,tp,dtype,tune", [("llama_tp4", 2, "bf16", ""), ("llama_tp4", 4, "bf16", ""), ("qwen2_a", 2, "f32", ""),
                                                ("llama_tp4", 2, "bf16", "gemv_blocks=6,gemv_waves=4"),
                                                ("llama_tp8", 8, "bf16", "gemv_blocks=16,gemv_waves=4"),
                                                ("mistral_wide", 2, "bf16", "gemv_blocks=96,gemv_waves=4")])
def test_all_reduce_fused_into_gemv_epilogue(tmp_path, name, tp, dtype, tune, monkeypatch):
    """Decode steps of a connected group exchange o_proj / down_proj partial sums in the GEMV epilogue (comm_ll.h)
    instead of a one-shot kernel per all-reduce.  Same addends in the same rank order: greedy AND sampled tokens
    (which follow the logits bit for bit) equal the unfused group's and the emulated group's, on every rank."""
    import fastllm_amd as fa
    from fastllm_amd import binding
    cfg = synth.CONFIGS[name]
    T, n_s = 12, 40
    (tmp_path / "fused").mkdir(); (tmp_path / "plain").mkdir()
    # Eight ranks on ONE card: a waiting wave needs its peers' waves resident, and only the GEMV grids can be cut (gemv_blocks).  The
    # short-cache form of the o_proj launch (k_attn_rep.hip) is one row per wave -- 128 workgroups x 8 ranks here, more than the card
    # holds (found by this test's first run: every rank's resident waves waited for rows of workgroups not yet dispatched, bounded
    # wait, FL_ERR_RCCL) -- so the 8-rank group keeps attention and o_proj apart; with a GPU per rank the grid is that GPU's alone.
    rep = {"FL_ATTN_REP": "0"} if tp > 4 else {}
    fused = run_group(tmp_path / "fused", name, dtype, tp, T, 2, 24, env_extra=dict(rep, TP_WORKER_SAMPLED=str(n_s), FL_TP_FUSED_AR="2", TP_WORKER_TUNE=tune))
    plain = run_group(tmp_path / "plain", name, dtype, tp, T, 2, 24, env_extra=dict(rep, TP_WORKER_SAMPLED=str(n_s), FL_TP_FUSED_AR="0", TP_WORKER_TUNE=tune))
    for r in range(tp):
        for k in ("tokens", "sampled", "decode"):
            np.testing.assert_array_equal(fused[r][k], fused[0][k], err_msg="fused rank %d vs 0: %s" % (r, k))
            np.testing.assert_array_equal(fused[r][k], plain[r][k], err_msg="fused vs one-shot kernel, rank %d: %s" % (r, k))
    w = synth.synth_weights(cfg)
    ids = synth.prompt_ids(cfg, T + 2, seed=11)
    # the workgroup size sets the order of the fused norm's sum of squares: compare like with like
    tuned = dict((k, int(v)) for k, v in (kv.split("=") for kv in filter(None, tune.split(","))))
    try:
        for k, v in tuned.items():
            fa.tune(k, v)
        if rep:
            monkeypatch.setenv("FL_ATTN_REP", "0")
        gE = fa.Model(cfg, w, dtype=dtype, tp_mode=binding.TP_EMULATED, tp_size=tp)
        cE = gE.new_cache(T + n_s + 8)
        f = gE.forward_sample(cE, ids[:T], 0, 0.9, seed=5)
        want = np.concatenate([[f], gE.decode_sample(cE, f, T, n_s, 0.9, seed=5)]).astype(np.uint32)
    finally:
        for k in tuned:
            fa.tune(k, 0)
    np.testing.assert_array_equal(fused[0]["sampled"], want)
    # one process, two shards on this device, per-shard graphs: the same exchange over plain peer pointers
    if tp == 2 and not tune:
        monkeypatch.setenv("FL_TP_FUSED_AR", "2")         # both shards on this one GPU: allowed because the grids are small
        gS = fa.Model(cfg, w, dtype=dtype, tp_mode=binding.TP_SINGLE_PROCESS, tp_size=tp, device_ids=[0] * tp)
        cS = gS.new_cache(T + n_s + 8)
        f = gS.forward_sample(cS, ids[:T], 0, 0.9, seed=5)
        np.testing.assert_array_equal(np.concatenate([[f], gS.decode_sample(cS, f, T, n_s, 0.9, seed=5)]).astype(np.uint32), want)
        gS.profile_begin()                                # ... and it really is the fused form: no all-reduce kernels in a step
        gS.decode_greedy(cS, f, T + n_s, 2)
        names = [k["name"] for k in gS.profile_end()]
        assert names and not any("allreduce" in n for n in names), names
        assert any("allgather" in n for n in names), names
        gS.close()
    gE.close()


@pytest.mark.parametrize("name,tp,fused", [("llama_a", 2, "2"), ("llama_a", 2, "0"), ("llama_tp8", 8, "2"), ("llama_tp8", 8, "0")])
def test_stalled_peer_is_an_error_not_a_hang(tmp_path, name, tp, fused):
    """A rank that never shows up for a decode step: the waiting ranks' polls (in the GEMV epilogue, or in the one-shot
    kernel) give up after FL_AR_TIMEOUT_MS, the step returns FL_ERR_RCCL, and every kernel of the step still drains."""
    tune = "gemv_blocks=16,gemv_waves=4" if tp > 2 else ""
    res = run_group(tmp_path, name, "bf16", tp, env_extra={"TP_WORKER_STALLED_RANK": str(tp - 1), "FL_AR_TIMEOUT_MS": "300", "FL_TP_FUSED_AR": fused,
                                                           "TP_WORKER_TUNE": tune, "FL_ATTN_REP": "0" if tp > 4 else "1"})
    for r in range(tp - 1):
        msg = str(res[r]["error"])
        assert "gave up waiting for a peer" in msg, msg
        assert ("0xa11e" in msg) == (fused == "2"), msg                   # which waiter reported: fused epilogue / one-shot kernel
        assert float(res[r]["waited_s"]) < 30.0


@pytest.mark.parametrize("tp", [2, 4])
def test_multiprocess_mid_prompt_kernels_at_7b_width(tmp_path, tp):
    """Round 4's prompt kernels (128 x 256 tiles with in-launch K slices and the RoPE epilogue, 224-column gate/up tiles, row scales
    as partial sums) had met the multi-process path only on toy widths.  Mistral-7B's layer shape, 512 tokens (BASELINE C3 / C4's
    prompt), ranks as separate processes: (a) the side-stream schedule (all-reduces of one row chunk under the other's GEMMs) -- all
    ranks the same bits, within bf16 tolerance of the fp32 oracle; (b) the same prompt with the all-reduces on the compute stream
    -- the emulated group's bits exactly; and the kernels that ran are the mid-prompt ones."""
    import fastllm_amd as fa
    from fastllm_amd import binding
    name, dtype, T = "mistral_wide", "bf16", 512
    cfg = synth.CONFIGS[name]
    res = run_group(tmp_path, name, dtype, tp, T=T, n_tf=2, n_greedy=4, env_extra={"TP_WORKER_PROFILE": "1", "FL_TP_OVERLAP": "1"})
    for r in range(1, tp):
        for k in ("prefill", "prefill_profiled", "decode", "tokens"):
            np.testing.assert_array_equal(res[r][k], res[0][k], err_msg="rank %d vs 0: %s" % (r, k))
    kernels = [str(k) for k in res[0]["prefill_kernels"]]
    assert any("h4," in k for k in kernels), kernels                    # 128 x 256 tiles, K slices met in the launch (k_gemm_h4.hip)
    w = synth.synth_weights(cfg)
    ids = synth.prompt_ids(cfg, T + 2, seed=11)
    gE = fa.Model(cfg, w, dtype=dtype, tp_mode=binding.TP_EMULATED, tp_size=tp)
    cE = gE.new_cache(T + 16)
    e = gE.forward(cE, ids[:T], 0)
    np.testing.assert_array_equal(res[0]["prefill_profiled"], e)
    om = oracle.OracleModel(cfg, synth.as_f32(w), round_bf16=True)
    oc = om.new_cache(T + 16)
    o = om.forward(oc, ids[:T], 0)
    check_logits(res[0]["prefill"], o, dtype, "multi-process overlapped prefill vs oracle")
    check_logits(e, o, dtype, "emulated prefill vs oracle")
    o1 = om.forward(oc, ids[T:T + 1], T)
    check_logits(res[0]["decode"][0], o1, dtype, "multi-process decode on the cache the prefill left vs oracle")
    gE.close()


@pytest.mark.parametrize("name,tp", [("llama_tp4", 2), ("llama_tp4", 4), ("llama_tp8", 8)])
def test_comm_selftest_proves_both_collective_forms(tmp_path, name, tp):
    """fl_comm_selftest -- the proof the RCCL bootstrap runs before it trusts the inboxes (comm_prove_oneshot: integer-valued data,
    exact sums, 2 s bounds) -- on groups wired by fl_comm_ipc_connect: 4096 floats (one workgroup), 16 384 and 98 304 (many), and
    the model is as usable afterwards as before (same bits as the emulated group)."""
    import fastllm_amd as fa
    from fastllm_amd import binding
    res = run_group(tmp_path, name, "bf16", tp, env_extra={"TP_WORKER_SELFTEST": "1", "FL_ATTN_REP": "0" if tp > 4 else "1"})
    for r in range(tp):
        assert res[r]["selftest"].tolist() == [True, True, True], "rank %d: %s" % (r, res[r]["selftest"])
    cfg = synth.CONFIGS[name]
    gE = fa.Model(cfg, synth.synth_weights(cfg), dtype="bf16", tp_mode=binding.TP_EMULATED, tp_size=tp)
    cE = gE.new_cache(32)
    np.testing.assert_array_equal(res[0]["prefill"], gE.forward(cE, synth.prompt_ids(cfg, 14, seed=11)[:10], 0))
    gE.close()


def test_continuous_batching_on_a_multiprocess_group(tmp_path):
    """fl_batch_replace on the ranks of a group: every rank swaps the same slot between the same two steps (a newcomer prefilled on the
    group), the batch goes on, every rank holds the same tokens -- and the newcomer's stream is what the group decodes for it alone."""
    import fastllm_amd as fa
    from fastllm_amd import binding
    name, tp, B = "llama_tp4", 4, 6
    res = run_group(tmp_path, name, "bf16", tp, env_extra={"TP_WORKER_BATCH": str(B), "TP_WORKER_BATCH_REPLACE": "1"})
    for r in range(1, tp):
        for k in ("batch_tokens", "batch_tokens_after_replace", "batch_replaced_first"):
            np.testing.assert_array_equal(res[r][k], res[0][k], err_msg="rank %d vs 0: %s" % (r, k))
    assert res[0]["batch_tokens_after_replace"].shape == (B, 6)
    cfg = synth.CONFIGS[name]
    gE = fa.Model(cfg, synth.synth_weights(cfg), dtype="bf16", tp_mode=binding.TP_EMULATED, tp_size=tp)
    cE = gE.new_cache(64)
    fE = gE.forward_argmax(cE, synth.prompt_ids(cfg, 9, seed=333), 0)
    assert int(res[0]["batch_replaced_first"]) == fE                       # the prefill is the group's single-sequence path: same bits
    alone = gE.decode_greedy(cE, fE, 9, 6)
    same = int(np.sum(res[0]["batch_tokens_after_replace"][1] == np.asarray(alone)))
    assert same >= 4, (res[0]["batch_tokens_after_replace"][1], alone)    # (batch step vs single step: another order of the fp32 sums; near-ties may flip late tokens)
    gE.close()


def test_stalled_peer_in_a_batch_step_is_an_error_not_a_hang(tmp_path):
    """... and the same for the many-workgroup collective behind a batch step's o_proj ([4, 4096] = 16 384 floats): every waiting
    workgroup gives up after FL_AR_TIMEOUT_MS, the last one still moves the epoch, the step returns FL_ERR_RCCL."""
    res = run_group(tmp_path, "mistral_wide", "bf16", 2, env_extra={"TP_WORKER_STALLED_RANK": "1", "FL_AR_TIMEOUT_MS": "300", "TP_WORKER_BATCH": "4"})
    msg = str(res[0]["error"])
    assert "gave up waiting for a peer" in msg and "0xa11d" in msg, msg
    assert float(res[0]["waited_s"]) < 30.0


@pytest.mark.parametrize("name,tp,B", [("llama_tp4", 2, 3), ("llama_tp4", 4, 12), ("mistral_wide", 2, 32), ("llama_tp8", 8, 9),
                                       ("mistral_wide", 4, 16), ("mistral_wide", 8, 8)])     # (the last three: [B, h] of 16384 floats and more -- the many-workgroup collective)
def test_batched_decode_on_a_multiprocess_group(tmp_path, name, tp, B):
    """Round 5: fl_batch_* on the ranks of an FL_TP_MULTI_PROCESS group -- the prefill-shaped step at T = B with a rank's shard shapes,
    one-shot all-reduces behind o_proj / down_proj (complete outputs, sums in rank order) and ONE gather of every rank's [B][V / tp]
    logits block.  Every rank holds the same bits; per sequence the logits are those of one GPU running the whole model in a batch
    (bf16 bar: the row-parallel sums run in another order), and the greedy loops agree up to near-ties."""
    import fastllm_amd as fa
    cfg = synth.CONFIGS[name]
    res = run_group(tmp_path, name, "bf16", tp, T=6, n_tf=1, n_greedy=2, env_extra={"TP_WORKER_BATCH": str(B), "FL_ATTN_REP": "0" if tp > 4 else "1"})
    for r in range(1, tp):
        for k in ("batch_logits", "batch_tokens"):
            np.testing.assert_array_equal(res[r][k], res[0][k], err_msg="rank %d vs 0: %s" % (r, k))
    w = synth.synth_weights(cfg)
    g1 = fa.Model(cfg, w, dtype="bf16")
    lens = [3 + (2 * i) % 11 for i in range(B)]
    caches, firsts = [], []
    for i, n in enumerate(lens):
        ci = g1.new_cache(48)
        firsts.append(g1.forward_argmax(ci, synth.prompt_ids(cfg, n, seed=70 + i), 0))
        caches.append(ci)
    feed = [int(t) for t in res[0]["batch_first"]]                   # (the group's own first tokens: a near-tie may have parted them)
    bt = fa.Batch(g1, caches)
    lg, am = bt.forward(feed, lens)
    same_first = sum(int(a == b) for a, b in zip(firsts, feed))
    assert same_first >= B - 1, (firsts, feed)
    for i in range(B):
        if firsts[i] == feed[i]:
            check_logits(res[0]["batch_logits"][i], lg[i], "bf16", "sequence %d: group vs one GPU" % i)
    assert res[0]["batch_tokens"].shape == (B, 11)
    bt.close()
    g1.close()
