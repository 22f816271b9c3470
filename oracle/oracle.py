"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module; nothing under fastllm_amd/ does (tests/test_layout.py enforces it).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

FAMILY = {"llama": 0, "mistral": 1, "qwen2": 2}
F32, BF16, F16 = 0, 1, 2


class OrcConfig(C.Structure):
    _fields_ = [("family", C.c_int32), ("qkv_bias", C.c_int32),
                ("hidden_size", C.c_int64), ("intermediate_size", C.c_int64),
                ("vocab_size", C.c_int64), ("num_hidden_layers", C.c_int64),
                ("num_attention_heads", C.c_int64), ("num_key_value_heads", C.c_int64),
                ("head_dim", C.c_int64), ("max_position_embeddings", C.c_int64),
                ("sliding_window", C.c_int64), ("rms_norm_eps", C.c_double),
                ("rope_theta", C.c_double)]


class OrcTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4), ("data", C.c_void_p)]


ALLREDUCE_FN = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.c_size_t, C.c_void_p)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("ref_forward.c", "ref_forward.h")]
    have_src = all(os.path.exists(s) for s in src)
    stale = have_src and os.path.exists(so) and os.path.getmtime(so) < max(os.path.getmtime(s) for s in src)
    if force or not os.path.exists(so) or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_model_create.argtypes = [C.POINTER(OrcConfig), C.POINTER(OrcTensor), C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
        L.orc_model_destroy.argtypes = [C.c_void_p]
        L.orc_model_destroy.restype = None
        L.orc_model_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
        L.orc_model_set_allreduce.restype = None
        L.orc_model_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.orc_model_set_threads.restype = None
        L.orc_model_threads.argtypes = [C.c_void_p]
        L.orc_cache_create.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        L.orc_cache_reset.argtypes = [C.c_void_p]
        L.orc_cache_reset.restype = None
        L.orc_cache_len.argtypes = [C.c_void_p]
        L.orc_cache_len.restype = C.c_size_t
        L.orc_cache_destroy.argtypes = [C.c_void_p]
        L.orc_cache_destroy.restype = None
        L.orc_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
        L.orc_argmax.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_argmax.restype = C.c_uint32
        L.orc_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int64, C.c_int,
                                   C.c_void_p, C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_chacha_block.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
        L.orc_chacha_block.restype = None
        L.orc_rng_seed_from_u64.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_rng_seed_from_u64.restype = None
        L.orc_rng_next_u32.argtypes = [C.c_void_p]
        L.orc_rng_next_u32.restype = C.c_uint32
        L.orc_sampler_init.argtypes = [C.c_void_p, C.c_uint64, C.c_double]
        L.orc_sampler_init.restype = None
        L.orc_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_sample.restype = C.c_uint32
        _LIB = L
    return _LIB


def default_threads(cap=16):
    """CPU threads this process may really use: affinity mask, cgroup quota, and at most `cap`
    (a one-GPU box shares its host: 16 CPUs per GPU)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


class OracleError(RuntimeError):
    pass


def _dtype_code(a):
    if a.dtype == np.float32:
        return F32
    if a.dtype == np.uint16:      # bf16 bit patterns
        return BF16
    if a.dtype == np.float16:
        return F16
    raise TypeError("unsupported array dtype %s" % a.dtype)


def make_config(cfg, head_dim=0):
    """cfg: dict with HF config.json keys + 'family'."""
    c = OrcConfig()
    c.family = FAMILY[cfg["family"]]
    c.qkv_bias = int(cfg.get("qkv_bias", cfg["family"] == "qwen2"))
    c.hidden_size = cfg["hidden_size"]
    c.intermediate_size = cfg["intermediate_size"]
    c.vocab_size = cfg["vocab_size"]
    c.num_hidden_layers = cfg["num_hidden_layers"]
    c.num_attention_heads = cfg["num_attention_heads"]
    c.num_key_value_heads = cfg.get("num_key_value_heads") or 0
    c.head_dim = head_dim
    c.max_position_embeddings = cfg.get("max_position_embeddings") or 0
    c.sliding_window = cfg.get("sliding_window") or 0
    c.rms_norm_eps = cfg["rms_norm_eps"]
    c.rope_theta = cfg.get("rope_theta") or 0.0
    return c


class OracleModel:
    def __init__(self, cfg, tensors, round_bf16=False, head_dim=0, threads=0):
        """tensors: dict name -> np.ndarray (float32, float16, or uint16 = bf16 bits)."""
        L = lib()
        self.cfg = dict(cfg)
        self.V = cfg["vocab_size"]
        arr = (OrcTensor * len(tensors))()
        keep = []
        for i, (name, a) in enumerate(tensors.items()):
            a = np.ascontiguousarray(a)
            keep.append(a)
            arr[i].name = name.encode()
            arr[i].dtype = _dtype_code(a)
            arr[i].ndim = a.ndim
            for j, s in enumerate(a.shape):
                arr[i].shape[j] = s
            arr[i].data = a.ctypes.data
        c = make_config(cfg, head_dim)
        h = C.c_void_p()
        if L.orc_model_create(C.byref(c), arr, len(tensors), int(round_bf16), C.byref(h)):
            raise OracleError(L.orc_last_error().decode())
        self._h = h
        self._cb = None
        L.orc_model_set_threads(h, threads or default_threads())

    def threads(self):
        return lib().orc_model_threads(self._h)

    def set_allreduce(self, fn):
        """fn(np.ndarray float32 view) -> None, reduces in place."""
        def tramp(ptr, n, _ctx):
            buf = np.ctypeslib.as_array(ptr, shape=(n,))
            fn(buf)
        self._cb = ALLREDUCE_FN(tramp)
        lib().orc_model_set_allreduce(self._h, self._cb, None)

    def new_cache(self, max_seq):
        return OracleCache(self, max_seq)

    def forward(self, cache, ids, pos):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty(self.V, dtype=np.float32)
        if lib().orc_forward(self._h, cache._h, ids.ctypes.data, ids.size, pos, out.ctypes.data):
            raise OracleError(lib().orc_last_error().decode())
        return out

    def generate(self, cache, prompt, max_tokens, eos=-1, pos_mode="tokens", want_logits=False):
        prompt = np.ascontiguousarray(prompt, dtype=np.uint32)
        toks = np.zeros(max_tokens, dtype=np.uint32)
        lg = np.zeros((max_tokens, self.V), dtype=np.float32) if want_logits else None
        n = lib().orc_generate(self._h, cache._h, prompt.ctypes.data, prompt.size, max_tokens, eos,
                               1 if pos_mode == "reference" else 0, toks.ctypes.data,
                               lg.ctypes.data if want_logits else None)
        if n < 0:
            raise OracleError(lib().orc_last_error().decode())
        return (toks[:n], lg[:n]) if want_logits else toks[:n]

    def close(self):
        if self._h:
            lib().orc_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OracleCache:
    def __init__(self, model, max_seq):
        h = C.c_void_p()
        if lib().orc_cache_create(model._h, max_seq, C.byref(h)):
            raise OracleError(lib().orc_last_error().decode())
        self._h = h
        self._model = model

    def reset(self):
        lib().orc_cache_reset(self._h)

    def __len__(self):
        return lib().orc_cache_len(self._h)

    def __del__(self):
        try:
            if self._h:
                lib().orc_cache_destroy(self._h)
                self._h = None
        except Exception:
            pass


def argmax(logits):
    a = np.ascontiguousarray(logits, dtype=np.float32)
    return int(lib().orc_argmax(a.ctypes.data, a.size))


class OrcRng(C.Structure):
    _fields_ = [("key", C.c_uint32 * 8), ("word", C.c_uint64)]


class OrcSamplerS(C.Structure):
    _fields_ = [("rng", OrcRng), ("temperature", C.c_double), ("argmax", C.c_int)]


class OrcSampleInfo(C.Structure):
    _fields_ = [("u", C.c_uint32), ("chosen", C.c_float), ("total", C.c_float), ("cum_lo", C.c_float),
                ("cum_hi", C.c_float), ("p", C.c_float)]


def chacha_block(key_words, counter, rounds):
    key = (C.c_uint32 * 8)(*key_words)
    out = (C.c_uint32 * 16)()
    lib().orc_chacha_block(key, counter, rounds, out)
    return list(out)


class Sampler:
    """LogitsProcessor::new(seed, temperature, None) of the reference's generate loop (ref_sampler.c).
    temperature None -> ArgMax."""

    def __init__(self, seed=0, temperature=None):
        self._s = OrcSamplerS()
        lib().orc_sampler_init(C.byref(self._s), seed, -1.0 if temperature is None else float(temperature))

    @property
    def key(self):
        return list(self._s.rng.key)

    @property
    def draws(self):
        return int(self._s.rng.word)

    def next_u32(self):
        return int(lib().orc_rng_next_u32(C.byref(self._s.rng)))

    def sample(self, logits, want_info=False):
        a = np.ascontiguousarray(logits, dtype=np.float32)
        scratch = np.empty_like(a)
        info = OrcSampleInfo()
        tok = int(lib().orc_sample(C.byref(self._s), a.ctypes.data, a.size, scratch.ctypes.data, C.byref(info)))
        if want_info:
            return tok, dict(u=info.u, chosen=info.chosen, total=info.total, cum_lo=info.cum_lo, cum_hi=info.cum_hi, p=info.p)
        return tok
