#!/usr/bin/env python3
"""bench.py -- decode tokens/s of the MI355X forward-pass backend on BASELINE.json's workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model mistral-7b] [--prompt 512]

One "step" = one pass of the hot path = ONE greedy decode forward (1 token, KV-cached).  The
default workload is BASELINE.json configs[2], the configuration the metric is quoted on:
Mistral-7B-v0.1 bf16, 512-token prompt, 256 generated tokens (K = 256 timed decode steps that
start right after the prompt).  N > 1 (launched by torch.distributed.run, one rank per GPU)
runs the same single sequence tensor-parallel over N GPUs (row/column weight shards + RCCL
all-reduce): total work is fixed, so scaling is "strong".

Timed region: inputs resident in HBM (weights, KV cache, prompt already prefilled); barrier +
device sync on both sides; max over ranks.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def synth_device_weights(torch, cfg, device, seed=1234, n_layers=None):
    """bf16 N(0, 0.02^2) weights generated directly in HBM (the reference loads safetensors straight
    onto the device, huggingface.rs:88,125).  Same seed on every rank => identical tensors."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import synth
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = {}
    cfg_l = dict(cfg)
    if n_layers is not None:
        cfg_l["num_hidden_layers"] = n_layers
    for name, shape in synth.tensor_shapes(cfg_l):
        t = torch.randn(shape, device=device, dtype=torch.float32, generator=g)
        if name.endswith("layernorm.weight") or name == "model.norm.weight":
            t = 1.0 + 0.02 * t
        elif name == "lm_head.weight":
            t = 0.16 * t              # x8: decisive argmax on random weights
        else:
            t = 0.02 * t
        out[name] = t.to(torch.bfloat16).contiguous()
    torch.cuda.synchronize(device)      # the library reads these on its own streams: they must be complete
    return out


def as_fl_tensors(tensors, dev_index):
    return {k: (v.data_ptr(), 1, tuple(v.shape), dev_index) for k, v in tensors.items()}   # 1 = FL_DTYPE_BF16


def cpu_baseline(torch, cfg, dev_tensors, kv_prompt=16, n_decode=8):
    """The reference's CPU path cannot be built here (Rust + candle, no toolchain); its stand-in is
    the C restatement in oracle/ ("port").  Bounded sample: the WHOLE model (every layer + lm_head, the same bf16
    weights copied to the host), n_decode greedy decode steps behind a kv_prompt-token prompt.  (Rounds 1-2
    extrapolated from 8 and 16 layers; the fit's intercept clamped to "head 0.0 ms" on some hosts.)"""
    from oracle import oracle
    threads = oracle.default_threads()
    prompt = np.arange(1, kv_prompt + 1, dtype=np.uint32)
    t0 = time.perf_counter()
    host = {k: v.view(torch.int16).cpu().numpy().view(np.uint16) for k, v in dev_tensors.items()}
    t_copy = time.perf_counter() - t0
    om = oracle.OracleModel(cfg, host, threads=threads)
    oc = om.new_cache(kv_prompt + n_decode + 2)
    tok = oracle.argmax(om.forward(oc, prompt, 0))
    tok = oracle.argmax(om.forward(oc, [tok], kv_prompt))             # one untimed step: first touch of every weight page
    t0 = time.perf_counter()
    for i in range(n_decode):
        tok = oracle.argmax(om.forward(oc, [tok], kv_prompt + 1 + i))
    t_tok = (time.perf_counter() - t0) / n_decode
    om.close()
    del host
    return {"value": round(1.0 / t_tok, 3), "unit": "tokens/s", "cores": threads, "kind": "port",
            "sample": "oracle/ref_forward.c (fp32 math on the same bf16 weights), all %d layers + lm_head, %d greedy decode steps "
                      "at kv_len %d..%d (a %d-token prompt, not the workload's 512: the CPU step is weight-bound, attention over "
                      "<= 1k keys is < 1 %% of it); %.1f ms per token; weights copied to the host in %.1f s (untimed)"
                      % (cfg["num_hidden_layers"], n_decode, kv_prompt + 1, kv_prompt + 1 + n_decode, kv_prompt, t_tok * 1e3, t_copy)}


def parity_check(torch, fa, binding, cfg, wts, local_rank, n_layers=4, T=512, n_decode=4):
    """Parity gate ("parity gate before any timing counts"): on the first n_layers layers + lm_head
    of the SAME synthetic weights, the bf16 HIP path (single GPU, through the C ABI) against oracle/ref_forward.c in fp32:
    last-position logits of a T-token prefill and of n_decode teacher-forced decode steps.  The oracle is the checker
    here, never the thing measured."""
    from oracle import oracle
    n_layers = min(n_layers, cfg["num_hidden_layers"])
    c2 = dict(cfg, num_hidden_layers=n_layers)
    sub = {k: v for k, v in wts.items() if not k.startswith("model.layers.") or int(k.split(".")[2]) < n_layers}
    host = {k: v.view(torch.int16).cpu().numpy().view(np.uint16) for k, v in sub.items()}
    om = oracle.OracleModel(c2, host, threads=oracle.default_threads())
    gm = fa.Model(c2, as_fl_tensors(sub, local_rank), dtype="bf16", tp_mode=binding.TP_NONE, device_ids=[local_rank])
    ids = np.random.RandomState(99).randint(0, cfg["vocab_size"], size=T + n_decode).astype(np.uint32)
    gc, oc = gm.new_cache(T + n_decode + 8), om.new_cache(T + n_decode + 8)
    max_abs, num, den, argmax_equal, margin_ok = 0.0, 0.0, 0.0, True, True
    for i in range(n_decode + 1):
        sl = ids[:T] if i == 0 else ids[T + i - 1:T + i]
        pos = 0 if i == 0 else T + i - 1
        lg, lo = gm.forward(gc, sl, pos), om.forward(oc, sl, pos)
        d = np.abs(lg - lo)
        max_abs = max(max_abs, float(d.max()))
        num += float(np.sum((lg - lo).astype(np.float64) ** 2)); den += float(np.sum(lo.astype(np.float64) ** 2))
        tg, to = int(np.flatnonzero(lg == lg.max())[-1]), int(oracle.argmax(lo))
        if tg != to:
            argmax_equal = False
            # bf16 may legitimately flip an argmax only where the fp32 reference itself is undecided at bf16 resolution
            margin_ok = margin_ok and float(lo[to] - lo[tg]) <= 2.0 * float(d.max())
    gc.close(); gm.close(); om.close()
    rel = (num / max(den, 1e-30)) ** 0.5
    ok = rel <= 2e-2 and margin_ok
    return {"ok": bool(ok), "max_abs": round(max_abs, 5), "rel_l2": round(rel, 6), "argmax_equal": bool(argmax_equal),
            "tolerance": "rel_l2 <= 2e-2 (bf16 path vs the fp32 oracle), argmax equal or inside 2*max_abs of the oracle's top",
            "sample": "first %d of %d layers + lm_head, %d-token prefill + %d decode steps, single GPU" % (n_layers, cfg["num_hidden_layers"], T, n_decode)}


def stream_batcher_leg(model, cfg, rs, slots=32, requests=64, prompt=128, gen=128):
    """fastllm::StreamBatcher (fastllm_amd/host) on the bench's own model handle: wall time from the first submit to the last token."""
    import ctypes as C
    host = C.CDLL(os.path.join(ROOT, "fastllm_amd", "lib", "libfastllm_host.so"))
    host.flh_last_error.restype = C.c_char_p
    TOK = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_uint32, C.c_void_p)
    DONE = C.CFUNCTYPE(None, C.c_uint64, C.c_size_t, C.c_void_p)
    host.flh_batcher_create_on.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    host.flh_batcher_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.c_int64, TOK, DONE, C.c_void_p, C.POINTER(C.c_uint64)]
    host.flh_batcher_run.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    host.flh_batcher_destroy.argtypes = [C.c_void_p]
    count = [0]

    def on_token(_rid, _tok, _u):
        count[0] += 1
        return 1
    cb, dcb = TOK(on_token), DONE(lambda _rid, _n, _u: None)
    prompts = [rs.randint(0, cfg["vocab_size"], size=prompt).astype(np.uint32) for _ in range(requests)]
    best = None
    for _rep in range(2):                                    # (the first pass captures the step's graph)
        b = C.c_void_p()
        if host.flh_batcher_create_on(model._h, slots, prompt + gen + 16, 16, 0, C.byref(b)) != 0:
            raise RuntimeError(host.flh_last_error().decode(errors="replace"))
        count[0] = 0
        t0 = time.perf_counter()
        for p in prompts:
            rid = C.c_uint64(0)
            if host.flh_batcher_submit(b, p.ctypes.data, p.size, gen, 0.0, -1, cb, dcb, None, C.byref(rid)) != 0:
                raise RuntimeError(host.flh_last_error().decode(errors="replace"))
        steps, pre = C.c_size_t(0), C.c_size_t(0)
        rc = host.flh_batcher_run(b, C.byref(steps), C.byref(pre))
        dt = time.perf_counter() - t0
        host.flh_batcher_destroy(b)
        if rc != 0:
            raise RuntimeError(host.flh_last_error().decode(errors="replace"))
        best = {"requests": requests, "slots": slots, "prompt_tokens": prompt, "generated_per_request": gen, "tokens": count[0], "seconds": round(dt, 4),
                "tokens_per_sec": round(count[0] / dt, 1), "batch_steps": steps.value, "prefills": pre.value,
                "note": "fastllm::StreamBatcher (host mirror, C++): requests queue for a slot of an fl_batch, are prefilled alone and decode together "
                        "(fl_batch_decode_each, 16 steps per call); wall time from the first submit to the last token, prefills included"}
    return best


def secondary_entry(torch, fa, binding, device, local_rank, name, T, gen, dtype="bf16", parity_layers=4, steps=None, with_parity=True):
    """One of the other single-GPU BASELINE configs (or the headline workload in the fp32 parity mode), measured after the headline
    with the same rules: its own parity gate against the oracle first, synthetic weights of the model's real shapes, the median of
    three prefills, `steps` greedy decode steps behind the prompt timed between device syncs."""
    from fastllm_amd.configs import MODEL_CONFIGS, decode_bytes_per_token, prefill_flops
    cfg = MODEL_CONFIGS[name]
    steps = steps or gen
    wts = synth_device_weights(torch, cfg, device)
    par = parity_check(torch, fa, binding, cfg, wts, local_rank, n_layers=parity_layers, T=min(T, 512), n_decode=4) if with_parity else None
    m = fa.Model(cfg, as_fl_tensors(wts, local_rank), dtype=dtype, tp_mode=binding.TP_NONE, device_ids=[local_rank])
    del wts
    torch.cuda.empty_cache()
    try:
        rs = np.random.RandomState(1234)
        prompt = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
        prompt[0] = 1
        cache = m.new_cache(T + max(steps, 8) + 72)
        first = m.forward_argmax(cache, prompt, 0)
        m.decode_greedy(cache, first, T, 8)                          # warm-up + graph capture
        samples = []
        for _ in range(3):
            cache.reset()
            m.synchronize()
            t0 = time.perf_counter()
            first = m.forward_argmax(cache, prompt, 0)
            m.synchronize()
            samples.append(time.perf_counter() - t0)
        t_prefill = sorted(samples)[1]
        m.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = m.decode_greedy(cache, first, T, steps)
        m.synchronize(); torch.cuda.synchronize()
        el = time.perf_counter() - t0
        assert len(toks) == steps
        tok_s = steps / el
        b_tok = decode_bytes_per_token(cfg, T + steps // 2, bytes_per_elem=2 if dtype == "bf16" else 4)
        cache.close()
        return {"workload": "%s %s greedy decode, %d-token prompt, %d generated tokens (%d timed), batch 1, TP=1" % (name, dtype, T, gen, steps),
                "tokens_per_sec": round(tok_s, 2), "ms_per_step": round(el / steps * 1e3, 4),
                "e2e_frac": round(tok_s * b_tok / 8e12, 4), "prefill_ms": round(t_prefill * 1e3, 2),
                "prefill_tokens_per_sec": round(T / t_prefill, 1),
                "mfma_frac": round(prefill_flops(cfg, T) / t_prefill / 2.5e15, 4) if dtype == "bf16" else None,
                "parity_check": par, "tokens_crc32": zlib.crc32(np.asarray(toks, dtype=np.uint32).tobytes())}
    finally:
        m.close()


def traffic_live(model_name, prompt, timeout_s=240):
    """HBM bytes per launch of the weight-streaming kernel, measured NOW: two child runs of this bench (16 decode steps, no CPU baseline,
    no secondary legs) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, counters alone with --kernel-trace, as
    /opt/skills/guides/MI355X_MICROARCH.md prescribes -- and the guide's gfx950 corrections (tools/pmc_traffic.py: KiB units, wide reads
    reported at half).  Children of this process (never exec); returns None if the profiler is not there or a pass fails."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    # never under a profiler already: its preloaded tool library initialises the GPU in every process it starts, and the inner
    # profiler's launcher would then replace itself (exec) with that state -- the boxes refuse exactly that
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        log("live HBM traffic passes skipped: this process already runs under a profiler")
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic
    out = tempfile.mkdtemp(prefix="fl_traffic_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", FL_BENCH_BATCH="0")
    try:
        res = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(out, ctr.lower())
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "16", "--warmup", "4", "--model", model_name, "--prompt", str(prompt), "--no-cpu-baseline", "--no-secondary", "--no-traffic"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            if r.returncode != 0:
                log("traffic pass %s failed (rc %d): %s" % (ctr, r.returncode, r.stderr.decode(errors="replace")[-300:]))
                return None
            res[ctr] = pmc_traffic.per_kernel(d, ctr, "gemv_kernel")
        (nf, fetch), (nw, write) = res["FETCH_SIZE"], res["WRITE_SIZE"]
        if not nf or not nw:
            return None
        rd, wr = 2.0 * 1024.0 * fetch / nf, 1024.0 * write / nw
        return {"hbm_bytes_per_launch": rd + wr, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "launches": nf}
    except Exception as e:                                   # noqa: the headline must not depend on the profiler
        log("live traffic measurement failed: %r" % (e,))
        return None
    finally:
        shutil.rmtree(out, ignore_errors=True)


def self_launch(n, argv):
    """`python bench.py --gpus N` run directly: start `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a CHILD process (never exec: under `rocprofv3 -- python3 bench.py` the profiler has already initialised the GPU
    in this process), relay rank 0's single JSON line and return the child's exit code (non-zero if any rank failed)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL and the peer inboxes need it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("[bench] self-launch:", " ".join(cmd), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True)
    try:
        out, _ = child.communicate(timeout=float(os.environ.get("FL_BENCH_TIMEOUT_S", "1500")))
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(child.pid, signal.SIGKILL)             # the exact process group started above
        out, _ = child.communicate()
        print("[bench] the %d-rank run exceeded FL_BENCH_TIMEOUT_S and was killed" % n, file=sys.stderr)
    line = None
    for ln in out.decode(errors="replace").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    if child.returncode != 0:
        print("[bench] the %d-rank run failed (exit code %d)" % (n, child.returncode), file=sys.stderr)
        return child.returncode if child.returncode > 0 else 1
    return 0 if line is not None else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default=os.environ.get("FL_BENCH_MODEL", "mistral-7b"))
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other single-GPU configs and the fp32-mode figure")
    ap.add_argument("--no-traffic", action="store_true", help="skip the in-run rocprofv3 PMC passes behind roofline.traffic")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # invoked directly (`python bench.py --gpus N`): start the ranks ourselves, BEFORE torch or the GPU are touched
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        args.gpus = world

    # stdout carries exactly ONE JSON line: native libraries (RCCL prints a banner at communicator
    # creation) are pointed at stderr for the duration of the run
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    import fastllm_amd as fa
    from fastllm_amd import binding
    from fastllm_amd.configs import MODEL_CONFIGS, decode_bytes_per_token, prefill_flops

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU path in the product)")
    # FL_BENCH_SAME_DEVICE=1: rehearsal of the N-rank path on a one-GPU box -- every rank drives cuda:0 with
    # 1/N of the weights.  RCCL refuses two ranks on one device, so the group is wired with the IPC
    # inboxes alone.  Ranks share one HBM: the tokens/s of such a run is NOT a scaling figure.
    same_device = world > 1 and os.environ.get("FL_BENCH_SAME_DEVICE", "0") == "1"
    if same_device:
        local_rank = 0
        # two processes with two live HIP queues each on ONE card make every launch of both ~25 us slower (measured:
        # profiles/r01/README.md); that is an artefact of the rehearsal, so keep the prefill's side stream out of it
        os.environ.setdefault("FL_TP_OVERLAP", "0")
        # the all-reduce fused into the GEMV epilogues spins inside full-chip grids: N ranks on one card would wait
        # for each other's workgroups to leave.  FL_BENCH_SAME_DEVICE_FUSED=1 rehearses it anyway at full model
        # size with the GEMV grids cut to 1/N of the card (see below); the default rehearsal keeps the kernel form.
        os.environ.setdefault("FL_TP_FUSED_AR", "2" if os.environ.get("FL_BENCH_SAME_DEVICE_FUSED", "0") == "1" else "0")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (unique-id broadcast, barriers, max-reduce of the time); the data path's
        # collectives are the library's own RCCL communicator over xGMI
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    cfg = MODEL_CONFIGS[args.model]
    T, K, W = args.prompt, args.steps, args.warmup
    log("model", args.model, "prompt", T, "steps", K, "warmup", W, "gpus", world)
    t0 = time.perf_counter()
    wts = synth_device_weights(torch, cfg, device)
    torch.cuda.synchronize()
    log("synthetic weights in HBM: %.1f GB in %.1fs" % (sum(v.numel() for v in wts.values()) * 2 / 1e9, time.perf_counter() - t0))

    def build_model():
        uid = None
        if world > 1 and not same_device:
            box = [fa.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
        mdl = fa.Model(cfg, as_fl_tensors(wts, local_rank), dtype="bf16",
                       tp_mode=binding.TP_MULTI_PROCESS if world > 1 else binding.TP_NONE, tp_size=world, tp_rank=rank,
                       device_ids=[local_rank], unique_id=uid)
        if same_device:
            hs = [None] * world
            dist.all_gather_object(hs, mdl.ipc_export())
            mdl.ipc_connect(hs)
        return mdl

    if same_device and os.environ.get("FL_BENCH_SAME_DEVICE_FUSED", "0") == "1":      # (FL_TP_FUSED_AR=0 on top: same grids, kernel form)
        fa.tune("gemv_blocks", 192 // world)
        fa.tune("gemv_waves", 4)
    t0 = time.perf_counter()
    # A tensor-parallel group is checked before it is measured: one overlapped prefill and a few decode steps; if any
    # rank fails (a bounded wait gave up, an RCCL error) ALL ranks rebuild the model one level more conservative:
    #   0 as configured;  1 all-reduces as kernels of their own, prefill all-reduces on the compute stream;  2 RCCL only.
    # The level that ran is reported in the JSON line (config.tp_fallback_level).
    os.environ.setdefault("FL_AR_TIMEOUT_MS", "5000")
    fallback_level, model = 0, None
    fallback_log = []                                # why a level was abandoned: (level, rank, error) of every rank that failed it
    for level in range(int(os.environ.get("FL_BENCH_START_LEVEL", "0")), 3):      # (tests start at a later level)
        if level >= 1:
            os.environ["FL_TP_FUSED_AR"] = "0"
            os.environ["FL_TP_OVERLAP"] = "0"
        if level >= 2:
            os.environ["FL_ONESHOT"] = "0"
        if level >= 1:
            fa.reload_env()                              # the library reads its switches once: have them re-read
        model = build_model()
        if world == 1:
            break
        failed, hc, why = 0, None, None
        try:
            hp = np.random.RandomState(7).randint(0, cfg["vocab_size"], size=max(T, 8)).astype(np.uint32)
            hc = model.new_cache(len(hp) + 16)
            if level == 0 and rank == world - 1 and os.environ.get("FL_BENCH_INJECT_LEVEL0_FAILURE") == "1":
                raise RuntimeError("injected health-check failure (test of the fallback path)")
            hf = model.forward_argmax(hc, hp, 0)
            model.decode_greedy(hc, hf, len(hp), 4)
            model.synchronize()
        except Exception as e:                                                # noqa: the point is to survive it
            failed, why = 1, repr(e)
            log("rank %d: tensor-parallel health check failed at level %d: %r" % (rank, level, e))
        finally:
            if hc is not None:
                hc.close()
        ft = torch.tensor([failed], dtype=torch.int32)
        dist.all_reduce(ft, op=dist.ReduceOp.MAX)
        if int(ft[0]) != 0:
            whys = [None] * world
            dist.all_gather_object(whys, why)
            fallback_log += [{"level": level, "rank": r, "error": w} for r, w in enumerate(whys) if w]
        if int(ft[0]) == 0:
            fallback_level = level
            break
        if level == 2:
            raise SystemExit("tensor-parallel group does not work even on the most conservative path")
        try:
            model.close()
        except Exception as e:
            log("rank %d: closing the failed model: %r" % (rank, e))
        model = None
    log("model built in %.1fs (tensor-parallel fallback level %d)" % (time.perf_counter() - t0, fallback_level))
    info = model.info()
    collectives = {0: None, 1: "rccl", 2: "one-shot peer inboxes (k_comm.hip) + rccl for prefill", 3: "local"}[info.small_collectives]
    if same_device:
        collectives = "one-shot peer inboxes (k_comm.hip); all ranks on ONE GPU (rehearsal)"
    if info.fused_all_reduce:
        collectives += "; decode all-reduces fused into the o_proj/down_proj GEMV epilogues (comm_ll.h)"
    log("decode collectives:", collectives)
    # the decode step's all-reduce ([h] fp32 after o_proj / down_proj) timed alone on this group's links, one form at a time
    # (fl_comm_probe; max over ranks): what a step's 2 L collectives cost by the form that carries them
    allreduce_us = None
    if world > 1:
        allreduce_us = {}
        for form, name in ((0, "rccl"), (1, "oneshot_kernel"), (2, "fused_in_gemv_epilogue")):
            try:
                us = model.comm_probe(form, cfg["hidden_size"], 32)
            except Exception as e:                                            # noqa: a probe must not take the bench down
                log("rank %d: comm_probe(%s) failed: %r" % (rank, name, e))
                us = None
            vals = [None] * world
            dist.all_gather_object(vals, us)
            allreduce_us[name] = None if any(v is None for v in vals) else round(max(vals), 2)
        allreduce_us["n_floats"] = cfg["hidden_size"]
        allreduce_us["per_step"] = 2 * cfg["num_hidden_layers"]
        log("decode all-reduce, us per call:", allreduce_us)

    # ---- parity gate: no timing counts before the HIP path has matched the oracle on this box ----
    parity = None
    if rank == 0:
        t0 = time.perf_counter()
        parity = parity_check(torch, fa, binding, cfg, wts, local_rank, T=min(T, 512))      # the workload's own prompt length (oracle: ~1 s)
        log("parity gate: %s (rel_l2 %.2e, max_abs %.3g, argmax_equal %s) in %.1fs"
            % ("ok" if parity["ok"] else "FAILED", parity["rel_l2"], parity["max_abs"], parity["argmax_equal"], time.perf_counter() - t0))
    if world > 1:
        # (rank 0 was busy with the oracle: the others must not already sit in a collective with a bounded wait)
        barrier()
        # the tensor-parallel group against ONE GPU running the whole model: same weights, same prompt, logits of the last position
        hp = np.random.RandomState(11).randint(0, cfg["vocab_size"], size=32).astype(np.uint32)
        hc = model.new_cache(48)
        lt = model.forward(hc, hp, 0)
        hc.close()
        if rank == 0:
            sm = fa.Model(cfg, as_fl_tensors(wts, local_rank), dtype="bf16", tp_mode=binding.TP_NONE, device_ids=[local_rank])
            sc_ = sm.new_cache(48)
            ls = sm.forward(sc_, hp, 0)
            sc_.close(); sm.close()
            rel = float(np.linalg.norm(lt - ls) / max(np.linalg.norm(ls), 1e-30))
            parity["tp_vs_single_gpu_rel_l2"] = round(rel, 6)
            # a tensor-parallel sum runs in another order: bf16 roundings downstream flip and 32 layers carry them along
            # (Mistral-7B, 2 ranks: 1.7e-2; the same model against itself with another attention split count: ~2e-2);
            # a wrong shard or a lost all-reduce is O(1)
            parity["ok"] = bool(parity["ok"] and rel <= 5e-2)
            parity["tp_tolerance"] = "rel_l2 <= 5e-2 against one GPU running the whole model (bf16, summation order differs)"
            log("parity gate: TP=%d logits vs one GPU: rel_l2 %.2e" % (world, rel))
        pk = torch.tensor([1 if (parity is None or parity["ok"]) else 0], dtype=torch.int32)
        dist.all_reduce(pk, op=dist.ReduceOp.MIN)
        parity_ok = int(pk[0]) == 1
    else:
        parity_ok = parity["ok"]

    do_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    cpu = None
    if do_cpu:
        t0 = time.perf_counter()
        cpu = cpu_baseline(torch, cfg, wts)
        log("cpu baseline: %s tok/s on %d threads (%.1fs)" % (cpu["value"], cpu["cores"], time.perf_counter() - t0))
    del wts
    torch.cuda.empty_cache()

    rs = np.random.RandomState(1234)
    prompt = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
    prompt[0] = 1
    cache = model.new_cache(T + max(K, W) + 64)

    # ---- warm-up: prefill + W decode steps (also captures the decode graph) ----
    first = model.forward_argmax(cache, prompt, 0)
    if W:
        model.decode_greedy(cache, first, T, W)
    # ---- prefill timing (reported beside the headline): the median of three calls on a reset cache ----
    samples = []
    for _ in range(3):
        cache.reset()
        barrier(); model.synchronize()
        t0 = time.perf_counter()
        first = model.forward_argmax(cache, prompt, 0)
        model.synchronize(); barrier()
        samples.append(time.perf_counter() - t0)
    t_prefill = sorted(samples)[1]

    # ---- per-kernel breakdown of the same prefill (eager, HIP-event pair per launch) ----
    cache.reset()
    model.profile_begin()
    model.forward_argmax(cache, prompt, 0)
    pstats = model.profile_end()
    barrier()

    # ---- timed region: exactly K decode steps ----
    barrier(); model.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks = model.decode_greedy(cache, first, T, K)
    model.synchronize(); torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed, t_prefill], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, t_prefill = float(tt[0]), float(tt[1])
    assert len(toks) == K
    # every rank of a tensor-parallel group must have produced the same tokens (the sums run in rank order everywhere)
    my_crc = zlib.crc32(np.asarray(toks, dtype=np.uint32).tobytes())
    ranks_agree = True
    if world > 1:
        crcs = [None] * world
        dist.all_gather_object(crcs, my_crc)
        ranks_agree = all(c == crcs[0] for c in crcs)
        if not ranks_agree:
            log("WARNING: ranks disagree on the generated tokens:", crcs)

    # ---- the reference's loop shape for comparison: one fl_forward per token, V fp32 logits copied to the host
    #      and argmax there (mod.rs:421-452): PCIe-inclusive, never the headline value ----
    n_host = 32
    tok = int(toks[-1])
    barrier(); model.synchronize()
    t0 = time.perf_counter()
    for i in range(n_host):
        lg = model.forward(cache, [tok], T + K + i)
        tok = int(np.flatnonzero(lg == lg.max())[-1])
    model.synchronize(); barrier()
    t_host_loop = (time.perf_counter() - t0) / n_host

    # ---- per-kernel HIP-event timing of the same decode steps (eager, event pair per launch) ----
    n_prof = 8
    model.profile_begin()
    model.decode_greedy(cache, tok, T + K + n_host, n_prof)
    stats = model.profile_end()
    barrier()

    # ---- batched decode (scope row N4: concurrent streams share one weight read); NOT the headline metric ----
    batch8, batch32, stream_batcher = None, None, None
    if world == 1 and os.environ.get("FL_BENCH_BATCH", "1") == "1":
        try:
            nb, kb = 8, min(K, 64)
            bc, bfirst = [], []
            for i in range(nb):
                pi = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
                ci = model.new_cache(T + kb + 24)
                bfirst.append(model.forward_argmax(ci, pi, 0))
                bc.append(ci)
            bt = fa.Batch(model, bc)
            g = bt.decode(bfirst, [T] * nb, 8)                              # warm-up + graph capture
            model.synchronize()
            t0 = time.perf_counter()
            g = bt.decode([int(x[-1]) for x in g], [T + 8] * nb, kb)
            model.synchronize()
            tb = time.perf_counter() - t0
            batch8 = {"streams": nb, "steps": kb, "ms_per_step": round(tb / kb * 1e3, 4),
                      "aggregate_tokens_per_sec": round(nb * kb / tb, 1), "per_stream_tokens_per_sec": round(kb / tb, 1),
                      "note": "fl_batch_decode: 8 independent %d-token-prompt streams advanced together (k_gemv_batch.hip)" % T}
            log("batched decode x8: %.1f tokens/s aggregate" % batch8["aggregate_tokens_per_sec"])
            bt.close()
            # 32 streams (round 5: fl_batch takes up to 64): 24 more caches behind the same prompts' length
            nb2 = 32
            for i in range(nb, nb2):
                pi = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
                ci = model.new_cache(T + kb + 24)
                bfirst.append(model.forward_argmax(ci, pi, 0))
                bc.append(ci)
            for ci in bc[:nb]:                                               # the first eight decoded on: back to the prompt
                ci.reset()
            for i in range(nb):
                pi = rs.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
                bfirst[i] = model.forward_argmax(bc[i], pi, 0)
            bt = fa.Batch(model, bc)
            g = bt.decode(bfirst, [T] * nb2, 8)
            model.synchronize()
            t0 = time.perf_counter()
            g = bt.decode([int(x[-1]) for x in g], [T + 8] * nb2, kb)
            model.synchronize()
            tb = time.perf_counter() - t0
            bb = decode_bytes_per_token(cfg, 0) + nb2 * (decode_bytes_per_token(cfg, T + 8 + kb // 2) - decode_bytes_per_token(cfg, 0))
            batch32 = {"streams": nb2, "steps": kb, "ms_per_step": round(tb / kb * 1e3, 4),
                       "aggregate_tokens_per_sec": round(nb2 * kb / tb, 1), "per_stream_tokens_per_sec": round(kb / tb, 1),
                       "hbm_frac_of_8TBps": round(bb / (tb / kb) / 8e12, 4),
                       "note": "fl_batch_decode: 32 independent %d-token-prompt streams advanced together: the prefill-shaped step at T = 32 "
                               "(short-prompt GEMM, batch attention); hbm_frac = (weights once + 32 streams' K / V) per step / 8 TB/s" % T}
            log("batched decode x32: %.1f tokens/s aggregate" % batch32["aggregate_tokens_per_sec"])
            bt.close()
            for ci in bc:
                ci.close()
            # continuous batching end to end (round 5): the host mirror's StreamBatcher (fastllm_host.hpp) -- 64 requests of 128 + 128
            # tokens through 32 slots, every request prefilled alone into a free slot and decoded with the others; prefills included
            try:
                stream_batcher = stream_batcher_leg(model, cfg, rs)
                log("stream batcher: %.0f tokens/s end to end" % stream_batcher["tokens_per_sec"])
            except Exception as e:                                           # noqa: the headline line must not depend on it
                log("stream batcher leg failed: %r" % (e,))
                stream_batcher = {"error": repr(e)}
        except Exception as e:                                               # the headline line must not depend on it
            log("batched decode leg failed:", repr(e))

    # ---- the same 32 streams on the tensor-parallel group (round 5: fl_batch_* on the ranks of a multi-process group).  Every rank builds
    #      the same batch; the headline's numbers are complete at this point and a failure here only drops this entry ----
    if world > 1 and os.environ.get("FL_BENCH_BATCH", "1") == "1":
        ok_b, bt, bc = 1, None, []
        try:
            nb2, kb = 32, min(K, 32)
            rsb = np.random.RandomState(4321)
            bfirst = []
            for i in range(nb2):
                pi = rsb.randint(0, cfg["vocab_size"], size=T).astype(np.uint32)
                ci = model.new_cache(T + kb + 24)
                bfirst.append(model.forward_argmax(ci, pi, 0))
                bc.append(ci)
            bt = fa.Batch(model, bc)
            g = bt.decode(bfirst, [T] * nb2, 8)
            barrier(); model.synchronize()
            t0 = time.perf_counter()
            g = bt.decode([int(x[-1]) for x in g], [T + 8] * nb2, kb)
            model.synchronize(); barrier()
            tb = time.perf_counter() - t0
        except Exception as e:                                               # noqa: the headline line must not depend on it
            ok_b = 0
            log("rank %d: tensor-parallel batched decode leg failed: %r" % (rank, e))
        okt = torch.tensor([ok_b], dtype=torch.int32)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt[0]) == 1:
            tt = torch.tensor([tb], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tb = float(tt[0])
            crcs = [None] * world
            dist.all_gather_object(crcs, zlib.crc32(np.stack(g).astype(np.uint32).tobytes()))
            batch32 = {"streams": nb2, "steps": kb, "ms_per_step": round(tb / kb * 1e3, 4), "aggregate_tokens_per_sec": round(nb2 * kb / tb, 1),
                       "per_stream_tokens_per_sec": round(kb / tb, 1), "ranks_agree": all(c == crcs[0] for c in crcs),
                       "note": "fl_batch_decode on the %d ranks of the group: 32 independent %d-token-prompt streams, one-shot all-reduces of [32, h] behind "
                               "o_proj / down_proj, one gather of the ranks' logits blocks per step" % (world, T)}
            log("batched decode x32 on the group: %.1f tokens/s aggregate" % batch32["aggregate_tokens_per_sec"])
        try:
            if bt is not None:
                bt.close()
            for ci in bc:
                ci.close()
        except Exception as e:                                               # noqa
            log("rank %d: closing the batch: %r" % (rank, e))

    # ---- the other single-GPU BASELINE configs and the fp32 parity mode, each behind its own parity gate (N = 1, default model only):
    #      configs[1] TinyLlama-1.1B 128 / 128; configs[4]'s single-GPU half, Qwen2-7B 4096-token prefill + decode at S = 4097..;
    #      and the headline workload in fp32 mode, the mode that meets north_star's literal 1e-3 / bit-exact-ids bar.
    #      The headline's numbers are complete at this point: nothing below can take them down. ----
    secondary, fp32_mode = None, None
    if world == 1 and rank == 0 and args.model == "mistral-7b" and not args.no_secondary and os.environ.get("FL_BENCH_SECONDARY", "1") == "1":
        secondary = []
        for nm, t_, gen_, steps_ in (("tinyllama-1.1b", 128, 128, 128), ("qwen2-7b", 4096, 64, 64)):
            try:
                t0 = time.perf_counter()
                e = secondary_entry(torch, fa, binding, device, local_rank, nm, t_, gen_, steps=steps_)
                if not (e["parity_check"] and e["parity_check"]["ok"]):
                    e["tokens_per_sec"] = None; e["invalid"] = "parity gate failed"
                secondary.append(e)
                log("secondary %s: %s tokens/s, prefill %.2f ms (%.1fs)" % (nm, e["tokens_per_sec"], e["prefill_ms"], time.perf_counter() - t0))
            except Exception as ex:                                          # noqa: the headline line must not depend on it
                log("secondary %s failed: %r" % (nm, ex))
                secondary.append({"workload": nm, "error": repr(ex)})
        try:
            t0 = time.perf_counter()
            e = secondary_entry(torch, fa, binding, device, local_rank, args.model, T, K, dtype="f32", steps=min(K, 32), with_parity=False)
            fp32_mode = {"tokens_per_sec": e["tokens_per_sec"], "ms_per_step": e["ms_per_step"], "prefill_ms": e["prefill_ms"], "workload": e["workload"],
                         "note": "fp32 weights / activations / KV: the mode tests/test_gpu_literal_configs.py holds to logits within 1e-3 of the "
                                 "oracle and identical greedy ids over this very workload (512 / 256, all 32 layers); 4 bytes per weight"}
            log("fp32 mode: %s tokens/s (%.1fs)" % (e["tokens_per_sec"], time.perf_counter() - t0))
        except Exception as ex:                                              # noqa
            log("fp32-mode leg failed: %r" % (ex,))
            fp32_mode = {"error": repr(ex)}

    # every rank's own kernel times (a tensor-parallel step is as slow as its slowest rank's launches + collectives)
    per_rank = None
    if world > 1:
        mine = {"rank": rank, "kernel_us_per_step": round(sum(s_["total_ms"] for s_ in stats) * 1e3 / n_prof, 1),
                "kernels": [{"name": s_["name"], "launches_per_step": s_["launches"] / n_prof,
                             "us_per_launch": round(s_["total_ms"] * 1e3 / s_["launches"], 2)} for s_ in stats]}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if rank == 0:
        tok_s = K / elapsed
        kv_mid = T + K // 2
        b_tok = decode_bytes_per_token(cfg, kv_mid)
        gv = [s for s in stats if s["name"].startswith("gemv")]
        gemv = dict(launches=sum(s["launches"] for s in gv), total_ms=sum(s["total_ms"] for s in gv),
                    bytes=sum(s["bytes"] for s in gv)) if gv else None
        roof = None
        if gemv and gemv["launches"]:
            avg_ms = gemv["total_ms"] / gemv["launches"]
            ach = gemv["bytes"] / gemv["launches"] / (avg_ms * 1e-3) / 1e9
            # HBM bytes per launch from the PMC passes committed with the profiles (tools/pmc_traffic.py);
            # only comparable when the workload is the one they were collected on
            traffic, tsrc = None, None
            import glob
            live = None
            if world == 1 and not args.no_traffic and os.environ.get("FL_BENCH_TRAFFIC", "1") == "1":
                t0 = time.perf_counter()
                live = traffic_live(args.model, T)
                log("live HBM traffic passes: %s (%.1fs)" % ("%.1f MB per GEMV launch" % (live["hbm_bytes_per_launch"] / 1e6) if live else "not available", time.perf_counter() - t0))
            tfs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "traffic_gemv.json")))      # newest round's PMC passes
            if live:
                traffic = round(live["hbm_bytes_per_launch"])
                tsrc = ("measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of this bench as child processes (16 decode steps each, "
                        "%d launches; separate passes, gfx950 x2 read correction)" % live["launches"])
            elif tfs and args.model == "mistral-7b" and world == 1:
                tj = json.load(open(tfs[-1]))
                traffic = round(tj["hbm_bytes_per_launch"])
                tsrc = "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, separate runs, gfx950 x2 read correction)" % os.path.relpath(tfs[-1], ROOT)
            # the same figure as rocprofv3 saw it when the committed profile was collected (tools/rocprof_gemv.py)
            rocprof = None
            rfs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9]*", "rocprof_gemv.json")))
            if rfs and args.model == "mistral-7b" and world == 1:
                rj = json.load(open(rfs[-1]))
                rocprof = {"avg_launch_us": rj["avg_launch_us"], "achieved": rj["achieved_GBps"], "frac": rj["frac"],
                           "source": os.path.relpath(rfs[-1], ROOT) + " (rocprofv3 --kernel-trace --stats of bench.py, collected with the committed profile)"}
            roof = {"bound": "hbm", "kernel": "gemv_kernel<bf16> (QKV/O/gate-up/down/lm_head weight stream)",
                    "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                    "traffic": traffic, "traffic_source": tsrc, "launches_per_step": gemv["launches"] // n_prof,
                    "bytes_per_launch": round(gemv["bytes"] / gemv["launches"]), "avg_launch_us": round(avg_ms * 1e3, 2),
                    "rocprof": rocprof}
        total_prof_ms = sum(s["total_ms"] for s in stats)
        out = {
            "metric": "decode_tokens_per_sec", "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "%s bf16 greedy decode, %d-token prompt, %d generated tokens, batch 1, TP=%d"
                                   % (args.model, T, K, world), "kv_len": "%d..%d" % (T, T + K),
                       "parallelism": "tp%d" % world, "collectives": collectives, "tp_fallback_level": fallback_level,
                       "rccl_ranks": int(info.rccl_ranks), "tp_fallback_log": fallback_log},
            "allreduce_us": allreduce_us,
            "per_rank": per_rank,
            "roofline": roof,
            "cpu_baseline": cpu,
            "e2e_hbm": {"bytes_per_token": b_tok, "achieved_GBps": round(tok_s * b_tok / world / 1e9, 1),
                        "frac_of_8TBps_per_gpu": round(tok_s * b_tok / world / 8e12, 4)},
            "parity_check": parity,
            "tokens_crc32": my_crc, "ranks_agree": ranks_agree,
            "batched_decode": batch8,
            "batched_decode_32": batch32,
            "stream_batcher": stream_batcher,
            "secondary": secondary,
            "fp32_mode_tokens_per_sec": fp32_mode["tokens_per_sec"] if fp32_mode and "tokens_per_sec" in fp32_mode else None,
            "fp32_mode": fp32_mode,
            "host_loop": {"tokens_per_sec": round(1.0 / t_host_loop, 2),
                          "note": "fl_forward per token: logits (V fp32) to the host + host argmax, PCIe-inclusive"},
            "prefill": {"tokens": T, "tokens_per_sec": round(T / t_prefill, 1), "ms": round(t_prefill * 1e3, 2), "timing": "median of 3 calls",
                        "mfma_frac_of_2.5PF": round(prefill_flops(cfg, T) / t_prefill / world / 2.5e15, 4),
                        "kernels": [{"name": s["name"], "launches": s["launches"], "ms": round(s["total_ms"], 3),
                                     "TFLOPs": round(s["flops"] / s["total_ms"] / 1e9, 1) if s["total_ms"] and s["flops"] else None}
                                    for s in pstats]},
            "kernels": [{"name": s["name"], "launches_per_step": s["launches"] / n_prof,
                         "us_per_step": round(s["total_ms"] * 1e3 / n_prof, 2),
                         "us_per_launch": round(s["total_ms"] * 1e3 / s["launches"], 2),
                         "GBps": round(s["bytes"] / s["total_ms"] / 1e6, 1) if s["total_ms"] else None,
                         "share": round(s["total_ms"] / total_prof_ms, 4) if total_prof_ms else None} for s in stats],
            "hbm_allocated_gb": round(info.hbm_bytes_allocated / 1e9, 2),
        }
        if not (parity_ok and ranks_agree):
            # a fast kernel whose results differ from the reference's is not done: no number is reported
            out["value"] = None
            out["invalid"] = "parity gate failed" if not parity_ok else "ranks disagree on the generated tokens"
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()
    if not (parity_ok and ranks_agree):
        raise SystemExit(3)


if __name__ == "__main__":
    main()
