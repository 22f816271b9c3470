#!/usr/bin/env python3
"""Generate tests/golden/*.npz from HuggingFace transformers (fp32, eager attention).

Why HF and not the reference: the reference (Rust + candle ^0.8.2, un-vendored) cannot be
built or imported in this image (no cargo/rustc) and its tests hold no
golden logits.  HF transformers is an independent public implementation of the same three
architectures; these vectors pin the CPU restatement (oracle/ref_forward.c), which in
turn is the checker for the HIP path.  Run in the build container only:

    python tests/golden/make_golden.py

Each fixture holds inputs + expected outputs only (no weights: they are re-derived from
the seed by tests/synth.py and guarded by a sha256 digest).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import synth  # noqa: E402

from transformers import (LlamaConfig, LlamaForCausalLM, MistralConfig, MistralForCausalLM,  # noqa: E402
                          Qwen2Config, Qwen2ForCausalLM)


def hf_model(cfg, weights_f32):
    fam = cfg["family"]
    common = dict(hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                  num_hidden_layers=cfg["num_hidden_layers"], num_attention_heads=cfg["num_attention_heads"],
                  num_key_value_heads=cfg.get("num_key_value_heads") or cfg["num_attention_heads"],
                  vocab_size=cfg["vocab_size"], rms_norm_eps=cfg["rms_norm_eps"], rope_theta=cfg["rope_theta"],
                  max_position_embeddings=cfg["max_position_embeddings"], tie_word_embeddings=False,
                  attn_implementation="eager")
    if fam == "llama":
        m = LlamaForCausalLM(LlamaConfig(attention_bias=False, mlp_bias=False, **common))
    elif fam == "mistral":
        # candle window (j + W < i masked => W+1 keys)  ==  HF window W+1
        m = MistralForCausalLM(MistralConfig(sliding_window=cfg["sliding_window"] + 1, **common))
    else:
        m = Qwen2ForCausalLM(Qwen2Config(sliding_window=cfg["sliding_window"] + 1, use_sliding_window=True,
                                         max_window_layers=0, **common))
    sd = {k: torch.from_numpy(v.copy()) for k, v in weights_f32.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in k or "inv_freq" in k for k in missing), missing
    return m.float().eval()


@torch.no_grad()
def run_case(name, T, n_gen, ref_calls=0):
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    m = hf_model(cfg, synth.as_f32(w))
    ids = synth.prompt_ids(cfg, T)
    inp = torch.from_numpy(ids.astype(np.int64))[None]
    out = m(inp, use_cache=True)
    rec = dict(prefill_logits=out.logits[0, -1].numpy().astype(np.float32))
    # a shorter prefix as a second prefill vector (T//2 tokens)
    out_h = m(inp[:, : max(1, T // 2)], use_cache=False)
    rec["prefill_half_logits"] = out_h.logits[0, -1].numpy().astype(np.float32)

    # greedy decode, "tokens" positions (HF default = true token positions)
    if n_gen:
        past = out.past_key_values
        logits = out.logits[0, -1]
        toks, step_logits = [], []
        for _ in range(n_gen):
            tok = int(torch.argmax(logits))
            toks.append(tok)
            step_logits.append(logits.numpy().astype(np.float32))
            o = m(torch.tensor([[tok]]), past_key_values=past, use_cache=True)
            past = o.past_key_values
            logits = o.logits[0, -1]
        rec["gen_tokens"] = np.array(toks, dtype=np.uint32)
        rec["gen_logits"] = np.stack(step_logits)

    # "reference" positions (quirk C.1): decode call n is rotated as position n
    if ref_calls:
        out = m(inp, use_cache=True)
        past = out.past_key_values
        logits = out.logits[0, -1]
        toks, step_logits = [], []
        for n in range(1, ref_calls + 1):
            tok = int(torch.argmax(logits))
            toks.append(tok)
            step_logits.append(logits.numpy().astype(np.float32))
            o = m(torch.tensor([[tok]]), past_key_values=past, use_cache=True,
                  position_ids=torch.tensor([[n]]))
            past = o.past_key_values
            logits = o.logits[0, -1]
        rec["ref_tokens"] = np.array(toks, dtype=np.uint32)
        rec["ref_logits"] = np.stack(step_logits)

    meta = dict(name=name, config=cfg, T=T, n_gen=n_gen, ref_calls=ref_calls, seed=0xFA57,
                weights_sha256=synth.weights_digest(w), transformers=__import__("transformers").__version__,
                torch=torch.__version__)
    rec["prompt"] = ids
    rec["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print(name, "T", T, "gen", n_gen, "->", {k: v.shape for k, v in rec.items() if k != "meta"})


def _masked_last_logits(m, cfg, T, seq, pos):
    """Last-position logits of `seq` (prompt of T tokens + decode tokens) in ONE cache-free pass with an explicit 4-D mask:
    prompt rows see candle's prefill mask (causal, j + sliding_window >= i; mistral.rs / qwen.rs via App. A.5), decode rows the
    whole prefix (candle builds no mask at T = 1).  HF's own cached decode would apply the window to the cache, which the
    reference does not."""
    n, W = len(seq), cfg["sliding_window"]
    dt = next(m.parameters()).dtype
    mask = torch.full((1, 1, n, n), torch.finfo(dt).min, dtype=dt)
    for i in range(n):
        lo = 0 if i >= T else max(0, i - W)
        mask[0, 0, i, lo:i + 1] = 0
    o = m(torch.tensor([[int(t) for t in seq]]), attention_mask=mask, position_ids=torch.tensor([list(pos)]), use_cache=False)
    return o.logits[0, -1].float().numpy().astype(np.float32)


@torch.no_grad()
def run_case_windowed_decode(name, T, n_gen, ref_calls):
    """A fixture whose window bites during the prefill AND that decodes on (round 3: Qwen2, whose window the reference
    forces on, qwen.rs:49-52).  Same keys as run_case; every vector comes from _masked_last_logits."""
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    m = hf_model(cfg, synth.as_f32(w))
    m16 = hf_model(cfg, synth.as_f32(w)).to(torch.bfloat16)
    ids = synth.prompt_ids(cfg, T)
    rec = dict(prompt=ids)
    rec["prefill_logits"] = _masked_last_logits(m, cfg, T, ids, range(T))
    rec["prefill_half_logits"] = _masked_last_logits(m, cfg, T // 2, ids[:T // 2], range(T // 2))
    rec["hf_bf16_prefill_logits"] = _masked_last_logits(m16, cfg, T, ids, range(T))

    def decode(n, ref_positions, model, force=None):
        seq, pos, toks, rows = list(ids), list(range(T)), [], []
        for k in range(n):
            lg = _masked_last_logits(model, cfg, T, seq, pos)
            tok = int(np.argmax(lg)) if force is None else int(force[k])
            toks.append(tok); rows.append(lg)
            seq.append(tok); pos.append(k + 1 if ref_positions else T + k)      # quirk C.1: decode call n rotated as position n
        return np.array(toks, dtype=np.uint32), np.stack(rows)
    rec["gen_tokens"], rec["gen_logits"] = decode(n_gen, False, m)
    rec["ref_tokens"], rec["ref_logits"] = decode(ref_calls, True, m)
    rec["hf_bf16_gen_logits"] = decode(n_gen, False, m16, force=rec["gen_tokens"])[1]
    meta = dict(name=name, config=cfg, T=T, n_gen=n_gen, ref_calls=ref_calls, seed=0xFA57,
                weights_sha256=synth.weights_digest(w), transformers=__import__("transformers").__version__,
                torch=torch.__version__, method="cache-free passes with an explicit 4-D mask")
    rec["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **rec)
    print(name, "T", T, "gen", n_gen, "->", {k: v.shape for k, v in rec.items() if k != "meta"})


@torch.no_grad()
def add_bf16(name):
    """HF in bfloat16 (CPU, eager) on the same weights, prompt and -- teacher-forced -- the fp32 run's greedy tokens: how far
    a straightforward bf16 execution of the architecture lands from fp32.  The GPU bf16 path is held to 1.5x that distance
    (tests/test_gpu_parity_bf16.py).  Existing arrays of the fixture are left untouched; only hf_bf16_* keys are added."""
    path = os.path.join(HERE, name + ".npz")
    rec = dict(np.load(path))
    cfg = synth.CONFIGS[name]
    w = synth.synth_weights(cfg)
    assert synth.weights_digest(w) == json.loads(bytes(rec["meta"]).decode())["weights_sha256"]
    m = hf_model(cfg, synth.as_f32(w)).to(torch.bfloat16)
    inp = torch.from_numpy(rec["prompt"].astype(np.int64))[None]
    out = m(inp, use_cache=True)
    rec["hf_bf16_prefill_logits"] = out.logits[0, -1].float().numpy()
    if "gen_tokens" in rec:
        past, logits, steps = out.past_key_values, out.logits[0, -1], []
        for tok in rec["gen_tokens"]:
            steps.append(logits.float().numpy())
            o = m(torch.tensor([[int(tok)]]), past_key_values=past, use_cache=True)
            past, logits = o.past_key_values, o.logits[0, -1]
        rec["hf_bf16_gen_logits"] = np.stack(steps)
    np.savez_compressed(path, **rec)
    d = rec["hf_bf16_prefill_logits"] - rec["prefill_logits"]
    print(name, "hf bf16 vs fp32 prefill: rel L2 %.3e" % (np.linalg.norm(d) / np.linalg.norm(rec["prefill_logits"])))


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == "--qwen2-win":       # round 3: one new fixture, the committed ones stay byte-identical
        run_case_windowed_decode("qwen2_win", T=16, n_gen=8, ref_calls=8)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--head-dims":       # round 4: head_dim 100 / 96 / 48 fixtures, the committed ones stay byte-identical
        run_case("llama_d100", T=8, n_gen=16)
        run_case("qwen2_d96", T=8, n_gen=16, ref_calls=8)
        run_case("mistral_d48", T=8, n_gen=16, ref_calls=8)
        for n in ("llama_d100", "qwen2_d96", "mistral_d48"):
            add_bf16(n)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--add-bf16":        # round 2: extend the committed fixtures in place
        for n in ("llama_a", "llama_mha", "mistral_a", "mistral_win", "qwen2_a"):
            add_bf16(n)
        sys.exit(0)
    run_case("llama_a", T=8, n_gen=16)
    run_case("llama_mha", T=5, n_gen=8)
    run_case("mistral_a", T=8, n_gen=16, ref_calls=8)
    run_case("mistral_win", T=16, n_gen=0)      # prefill longer than the window; no decode
    run_case("qwen2_a", T=8, n_gen=16, ref_calls=8)
    for n in ("llama_a", "llama_mha", "mistral_a", "mistral_win", "qwen2_a"):
        add_bf16(n)
    run_case_windowed_decode("qwen2_win", T=16, n_gen=8, ref_calls=8)
