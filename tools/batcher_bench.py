"""fastllm::StreamBatcher at full size: R requests (P-token prompts, G generated tokens each, greedy) through S slots on Mistral-7B,
against the same requests served one after the other by flh_generate_stream (the reference's loop shape: one stream = one weight read per
token).  Wall time from the first submit to the last token, prefills included.  usage: batcher_bench.py [slots] [requests] [prompt] [gen]"""
import ctypes as C, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
from fastllm_amd import binding
from fastllm_amd.configs import MODEL_CONFIGS
S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
P = int(sys.argv[3]) if len(sys.argv) > 3 else 128
G = int(sys.argv[4]) if len(sys.argv) > 4 else 128
os.environ["FASTLLM_POS_MODE"] = "tokens"
os.environ["FASTLLM_MAX_SEQ"] = str(P + G + 16)
host = C.CDLL(os.path.join(ROOT, "fastllm_amd", "lib", "libfastllm_host.so"))
host.flh_last_error.restype = C.c_char_p
cfg = MODEL_CONFIGS["mistral-7b"]
wts = bench.synth_device_weights(torch, cfg, torch.device("cuda", 0))
arr = (binding.FlTensor * len(wts))()
for i, (k, v) in enumerate(wts.items()):
    arr[i].name, arr[i].dtype, arr[i].ndim, arr[i].data, arr[i].device = k.encode(), 1, v.dim(), v.data_ptr(), 0
    for j, s_ in enumerate(v.shape):
        arr[i].shape[j] = s_
d = {k: v for k, v in cfg.items() if k not in ("family", "qkv_bias") and v is not None}
d["architectures"] = ["MistralForCausalLM"]; d["torch_dtype"] = "bfloat16"
h = C.c_void_p()
host.flh_model_create.argtypes = [C.c_int, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
assert host.flh_model_create(1, json.dumps(d).encode(), arr, len(wts), 1, 0, C.byref(h)) == 0, host.flh_last_error()
del wts; torch.cuda.empty_cache()
rs = np.random.RandomState(0)
prompts = [rs.randint(0, cfg["vocab_size"], size=P).astype(np.uint32) for _ in range(R)]
TOK = C.CFUNCTYPE(C.c_int, C.c_uint64, C.c_uint32, C.c_void_p)
DONE = C.CFUNCTYPE(None, C.c_uint64, C.c_size_t, C.c_void_p)
count = [0]
def on_token(rid, tok, _u):
    count[0] += 1
    return 1
cb, dcb = TOK(on_token), DONE(lambda rid, n, _u: None)
host.flh_batcher_create.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
host.flh_batcher_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.c_int64, TOK, DONE, C.c_void_p, C.POINTER(C.c_uint64)]
host.flh_batcher_run.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
host.flh_batcher_destroy.argtypes = [C.c_void_p]
for rep in range(2):                                        # (first pass: graph capture, page faults)
    b = C.c_void_p()
    assert host.flh_batcher_create(h, S, P + G + 16, 16, C.byref(b)) == 0, host.flh_last_error()
    count[0] = 0
    t0 = time.perf_counter()
    for p in prompts:
        rid = C.c_uint64(0)
        assert host.flh_batcher_submit(b, p.ctypes.data, p.size, G, 0.0, -1, cb, dcb, None, C.byref(rid)) == 0, host.flh_last_error()
    steps, pre = C.c_size_t(0), C.c_size_t(0)
    assert host.flh_batcher_run(b, C.byref(steps), C.byref(pre)) == 0, host.flh_last_error()
    dt = time.perf_counter() - t0
    host.flh_batcher_destroy(b)
print("StreamBatcher, mistral-7b bf16: %d requests (%d-token prompts, %d tokens each) through %d slots: %.2f s, %d tokens -> %.0f tokens/s (%d batch steps, %d prefills)"
      % (R, P, G, S, dt, count[0], count[0] / dt, steps.value, pre.value), flush=True)
TOK1 = C.CFUNCTYPE(C.c_int, C.c_uint32, C.c_void_p)
c1 = [0]
def on1(tok, _u):
    c1[0] += 1
    return 1
cb1 = TOK1(on1)
host.flh_generate_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.c_int64, TOK1, C.c_void_p, C.POINTER(C.c_size_t)]
n1 = min(R, 8)
fw = C.c_size_t(0)
host.flh_generate_stream(h, prompts[0].ctypes.data, P, 8, 0.0, -1, cb1, None, C.byref(fw))
c1[0] = 0
t0 = time.perf_counter()
for p in prompts[:n1]:
    assert host.flh_generate_stream(h, p.ctypes.data, p.size, G, 0.0, -1, cb1, None, C.byref(fw)) == 0
dt1 = time.perf_counter() - t0
print("one stream after the other (flh_generate_stream, logits to the host per token): %d requests %.2f s -> %.0f tokens/s" % (n1, dt1, c1[0] / dt1))
