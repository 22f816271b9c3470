"""world_size-2 `gloo` rehearsal of the tensor-parallel path on CPU.

Each rank cuts its shard with the PRODUCT's partition function (fl_tp_slice: a pure host entry of the
C ABI), runs the shard through the oracle with an all-reduce(sum) hook after o_proj and down_proj --
the two collective call sites of the MI355X path (fastllm_amd/csrc/model.hip all_reduce_delta) -- and
the vocab-parallel logits are all-gathered.  TP=2 must reproduce TP=1.
"""
import os
import socket
import sys

import numpy as np
import pytest

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import fastllm_amd as fa
    from oracle import oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.CONFIGS[name]
    w = synth.as_f32(synth.synth_weights(cfg))
    H = cfg["num_attention_heads"]
    Hkv = cfg.get("num_key_value_heads") or H
    d = cfg["hidden_size"] // H
    shard = {}
    for k, a in w.items():
        r0, r1, c0, c1 = fa.tp_slice(cfg, k, rank, world)
        shard[k] = np.ascontiguousarray(a[r0:r1, c0:c1] if a.ndim == 2 else a[r0:r1])
    scfg = dict(cfg, num_attention_heads=H // world, num_key_value_heads=Hkv // world,
                intermediate_size=cfg["intermediate_size"] // world, vocab_size=cfg["vocab_size"] // world)
    # the embedding stays whole: look it up with the full vocab, so give the shard model the full table
    scfg_embed_rows = cfg["vocab_size"]
    shard["model.embed_tokens.weight"] = w["model.embed_tokens.weight"]
    # oracle shard: vocab_size is used for both embed and lm_head shapes, so pad lm_head rows to the full
    # vocab with zeros and cut the local slice of the logits afterwards
    lm = np.zeros_like(w["lm_head.weight"])
    r0, r1, _, _ = fa.tp_slice(cfg, "lm_head.weight", rank, world)
    lm[r0:r1] = w["lm_head.weight"][r0:r1]
    shard["lm_head.weight"] = lm
    scfg["vocab_size"] = scfg_embed_rows
    om = oracle.OracleModel(scfg, shard, head_dim=d, threads=2)

    def allreduce(buf):
        t = torch.from_numpy(buf)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    om.set_allreduce(allreduce)
    ids = synth.prompt_ids(cfg, 12, seed=21)
    oc = om.new_cache(32)
    outs = [om.forward(oc, ids[:9], 0)]
    for i in range(9, 12):
        outs.append(om.forward(oc, ids[i:i + 1], i))
    # vocab-parallel all-gather: every rank contributes its row slice
    full = []
    for lg in outs:
        part = torch.from_numpy(np.ascontiguousarray(lg[r0:r1]))
        parts = [torch.empty_like(part) for _ in range(world)]
        dist.all_gather(parts, part)
        full.append(torch.cat(parts).numpy())
    if rank == 0:
        np.save(os.path.join(outdir, "tp_logits.npy"), np.stack(full))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["mistral_a", "qwen2_a"])
def test_tp2_gloo_matches_tp1(name, tmp_path):
    import torch.multiprocessing as mp
    from oracle import oracle
    port = _free_port()
    mp.spawn(_worker, args=(2, port, name, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "tp_logits.npy"))
    cfg = synth.CONFIGS[name]
    om = oracle.OracleModel(cfg, synth.as_f32(synth.synth_weights(cfg)), threads=2)
    ids = synth.prompt_ids(cfg, 12, seed=21)
    oc = om.new_cache(32)
    ref = [om.forward(oc, ids[:9], 0)] + [om.forward(oc, ids[i:i + 1], i) for i in range(9, 12)]
    np.testing.assert_allclose(got, np.stack(ref), atol=2e-4, rtol=0)   # summation order changes only
