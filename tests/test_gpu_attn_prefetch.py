"""Prefetch workgroups in the decode attention launch (k_attn_mfma.hip, FL_ATTN_PREFETCH=1, off by default): extra rows of the grid
touch the next launch's o_proj weights while HBM idles.  They only read; the step's results are bit-identical with and without them,
at Mistral's grouping (8 kv heads: the grid is a multiple of eight as it stands) and at TinyLlama's (4 kv heads: rows are padded)."""
import numpy as np
import pytest

import synth
from test_gpu_fullsize import pooled_weights

from conftest import needs_experimental

pytestmark = [pytest.mark.gpu, needs_experimental]


@pytest.mark.parametrize("model,layers", [("tinyllama-1.1b", 3), ("mistral-7b", 2)])
def test_prefetch_workgroups_change_nothing_but_time(model, layers, monkeypatch):
    import fastllm_amd as fa
    from fastllm_amd.configs import MODEL_CONFIGS
    monkeypatch.setenv("FL_ATTN_REP", "0")                   # (short caches would otherwise run the replicated attention)
    cfg = dict(MODEL_CONFIGS[model], num_hidden_layers=layers)
    w = pooled_weights(cfg)
    ids = synth.prompt_ids(cfg, 200, seed=5)
    out = {}
    for mode, pct in (("0", "100"), ("1", "100"), ("1", "60")):
        monkeypatch.setenv("FL_ATTN_PREFETCH", mode)
        monkeypatch.setenv("FL_ATTN_PREFETCH_PCT", pct)
        gm = fa.Model(cfg, w, dtype="bf16")
        c = gm.new_cache(256)
        gm.forward(c, ids[:150], 0)
        out[mode + pct] = [gm.forward(c, ids[i:i + 1], i) for i in range(150, 170)]
        first = int(np.argmax(out[mode + pct][-1]))
        out[mode + pct].append(gm.decode_greedy(c, first, 170, 24))          # the graph path
        gm.close()
    for k in ("1100", "160"):
        for a, b in zip(out[k], out["0100"]):
            assert np.array_equal(a, b), "prefetch mode %s changed a result" % k
