"""Create / destroy models, caches and batches repeatedly and watch free HBM; long graph-replayed decode."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ctypes as C
import fastllm_amd as fa
import synth

hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")


def free_mb():
    f, t = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value / 2 ** 20


cfg = synth.CONFIGS["mistral_a"]
w = synth.synth_weights(cfg)
m0 = fa.Model(cfg, w, dtype="bf16")          # warm the runtime
c0 = m0.new_cache(64); m0.forward(c0, [1, 2, 3], 0); c0.close(); m0.close()
base = free_mb()
for it in range(6):
    m = fa.Model(cfg, w, dtype="bf16")
    caches = []
    for j in range(40):
        c = m.new_cache(200)
        f = m.forward_argmax(c, synth.prompt_ids(cfg, 5 + j % 7, seed=j), 0)
        m.decode_greedy(c, f, 5 + j % 7, 20)
        if j % 3 == 0:
            m.decode_sample(c, f, 25 + j % 7, 10, 0.8, draws_done=0)
        caches.append(c)
    for k in range(0, 40, 8):
        b = fa.Batch(m, caches[k:k + 8])
        b.decode([1] * 8, [len(c) for c in caches[k:k + 8]], 12)
        b.close()
    for c in caches:
        c.close()
    m.close()
    print("iteration %d: free HBM delta %.1f MB" % (it, free_mb() - base), flush=True)
assert abs(free_mb() - base) < 64, "HBM leak"
# long decode on one cache: 3000 graph-replayed steps, then the same from a fresh cache: identical ids
cfg = synth.CONFIGS["llama_a"]
w = synth.synth_weights(cfg)
m = fa.Model(dict(cfg, max_position_embeddings=8192), w, dtype="bf16")
ids = synth.prompt_ids(cfg, 9)
outs = []
for rep in range(2):
    c = m.new_cache(4096)
    f = m.forward_argmax(c, ids, 0)
    t0 = time.time()
    outs.append(m.decode_greedy(c, f, len(ids), 3000))
    print("3000 steps in %.2f s" % (time.time() - t0), flush=True)
assert np.array_equal(outs[0], outs[1]) and len(outs[0]) == 3000
print("soak ok")
