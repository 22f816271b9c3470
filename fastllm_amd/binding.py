"""ctypes binding of include/fastllm_mi355x.h (harness only; no arithmetic here).

Loading fails loudly when the HIP library has not been built: there is no fallback path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FL_LIB_PATH") or os.path.join(_HERE, "lib", "libfastllm_mi355x.so")   # FL_LIB_PATH: kernel-variant experiments

FAMILY = {"llama": 0, "mistral": 1, "qwen2": 2}
F32, BF16, F16 = 0, 1, 2
TP_NONE, TP_SINGLE_PROCESS, TP_MULTI_PROCESS, TP_EMULATED = 0, 1, 2, 3
UNIQUE_ID_BYTES = 128
IPC_HANDLE_BYTES = 64


class FastLLMError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fastllm_mi355x error %d: %s" % (code, msg))
        self.code = code


class FlConfig(C.Structure):
    _fields_ = [("family", C.c_int32), ("qkv_bias", C.c_int32), ("hidden_size", C.c_int64),
                ("intermediate_size", C.c_int64), ("vocab_size", C.c_int64), ("num_hidden_layers", C.c_int64),
                ("num_attention_heads", C.c_int64), ("num_key_value_heads", C.c_int64),
                ("max_position_embeddings", C.c_int64), ("sliding_window", C.c_int64),
                ("rms_norm_eps", C.c_double), ("rope_theta", C.c_double)]


class FlTensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("ndim", C.c_int32), ("shape", C.c_int64 * 4),
                ("data", C.c_void_p), ("device", C.c_int32), ("_pad", C.c_int32)]


class FlParallel(C.Structure):
    _fields_ = [("mode", C.c_int32), ("tp_size", C.c_int32), ("tp_rank", C.c_int32), ("n_device_ids", C.c_int32),
                ("device_ids", C.POINTER(C.c_int32)), ("unique_id", C.c_void_p)]


class FlModelInfo(C.Structure):
    _fields_ = [("cfg", FlConfig), ("head_dim", C.c_int64), ("compute_dtype", C.c_int32), ("tp_size", C.c_int32),
                ("weight_bytes_per_token", C.c_int64), ("kv_bytes_per_position", C.c_int64),
                ("hbm_bytes_allocated", C.c_int64), ("small_collectives", C.c_int32), ("fused_all_reduce", C.c_int32),
                ("rccl_ranks", C.c_int32), ("_reserved", C.c_int32)]


class FlSampling(C.Structure):
    _fields_ = [("temperature", C.c_double), ("seed", C.c_uint64), ("draws_done", C.c_uint64)]


class FlKernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double), ("bytes", C.c_double),
                ("flops", C.c_double)]


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, sz = C.c_void_p, C.c_size_t
        L.fl_abi_version.restype = C.c_int
        L.fl_last_error.restype = C.c_char_p
        L.fl_device_count.argtypes = [C.POINTER(C.c_int)]
        L.fl_comm_unique_id.argtypes = [vp]
        L.fl_comm_ipc_export.argtypes = [vp, vp]
        L.fl_comm_ipc_connect.argtypes = [vp, vp]
        L.fl_model_create.argtypes = [C.POINTER(FlConfig), C.POINTER(FlTensor), sz, C.c_int32, C.POINTER(FlParallel), C.POINTER(vp)]
        L.fl_model_retain.argtypes = [vp]
        L.fl_model_retain.restype = None
        L.fl_model_release.argtypes = [vp]
        L.fl_model_release.restype = None
        L.fl_model_get_info.argtypes = [vp, C.POINTER(FlModelInfo)]
        L.fl_cache_create.argtypes = [vp, sz, C.POINTER(vp)]
        L.fl_cache_reset.argtypes = [vp]
        L.fl_cache_reset.restype = None
        L.fl_cache_len.argtypes = [vp]
        L.fl_cache_len.restype = sz
        L.fl_cache_capacity.argtypes = [vp]
        L.fl_cache_capacity.restype = sz
        L.fl_cache_destroy.argtypes = [vp]
        L.fl_cache_destroy.restype = None
        L.fl_forward.argtypes = [vp, vp, vp, sz, sz, vp]
        L.fl_forward_argmax.argtypes = [vp, vp, vp, sz, sz, vp]
        L.fl_decode_greedy.argtypes = [vp, vp, C.c_uint32, sz, sz, C.c_int64, vp, C.POINTER(sz)]
        L.fl_forward_sample.argtypes = [vp, vp, vp, sz, sz, C.POINTER(FlSampling), vp]
        L.fl_decode_sample.argtypes = [vp, vp, C.c_uint32, sz, sz, C.c_int64, C.POINTER(FlSampling), vp, C.POINTER(sz)]
        L.fl_op_sample.argtypes = [vp, C.c_int64, C.POINTER(FlSampling), C.c_int64, vp]
        L.fl_op_attention.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp]
        L.fl_batch_create.argtypes = [vp, vp, sz, C.POINTER(vp)]
        L.fl_batch_destroy.argtypes = [vp]
        L.fl_batch_destroy.restype = None
        L.fl_batch_replace.argtypes = [vp, C.c_size_t, vp]
        L.fl_batch_decode_each.argtypes = [vp, vp, vp, C.c_size_t, vp, vp, vp, vp]
        L.fl_batch_forward.argtypes = [vp, vp, vp, vp, vp]
        L.fl_batch_decode.argtypes = [vp, vp, vp, sz, C.c_int64, C.POINTER(FlSampling), vp, vp]
        L.fl_synchronize.argtypes = [vp]
        L.fl_tp_slice.argtypes = [C.POINTER(FlConfig), C.c_char_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]
        L.fl_profile_begin.argtypes = [vp]
        L.fl_profile_end.argtypes = [vp, C.POINTER(FlKernelStat), sz, C.POINTER(sz)]
        L.fl_tune.argtypes = [C.c_char_p, C.c_int]
        L.fl_comm_probe.argtypes = [vp, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
        L.fl_comm_selftest.argtypes = [vp, C.c_int64, C.POINTER(C.c_int32)]
        L.fl_op_linear.argtypes = [vp, vp, vp, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, vp, C.c_int32,
                                   C.POINTER(C.c_double)]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise FastLLMError(rc, lib().fl_last_error().decode(errors="replace"))


def op_sample(logits, n_draws, temperature, seed=0, draws_done=0):
    """The token-selection kernel alone on a host logits vector: n_draws successive tokens."""
    a = np.ascontiguousarray(logits, dtype=np.float32)
    out = np.zeros(n_draws, dtype=np.uint32)
    sp = FlSampling(temperature, seed, draws_done)
    _check(lib().fl_op_sample(a.ctypes.data, a.size, C.byref(sp), n_draws, out.ctypes.data))
    return out


def abi_version():
    return lib().fl_abi_version()


def device_count():
    n = C.c_int(0)
    _check(lib().fl_device_count(C.byref(n)))
    return n.value


def comm_unique_id():
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    _check(lib().fl_comm_unique_id(buf))
    return buf.raw


def make_config(cfg):
    c = FlConfig()
    c.family = FAMILY[cfg["family"]] if isinstance(cfg["family"], str) else int(cfg["family"])
    c.qkv_bias = int(cfg.get("qkv_bias", cfg["family"] == "qwen2"))
    c.hidden_size = cfg["hidden_size"]
    c.intermediate_size = cfg["intermediate_size"]
    c.vocab_size = cfg["vocab_size"]
    c.num_hidden_layers = cfg["num_hidden_layers"]
    c.num_attention_heads = cfg["num_attention_heads"]
    c.num_key_value_heads = cfg.get("num_key_value_heads") or 0
    c.max_position_embeddings = cfg.get("max_position_embeddings") or 0
    c.sliding_window = cfg.get("sliding_window") or 0
    c.rms_norm_eps = cfg["rms_norm_eps"]
    c.rope_theta = cfg.get("rope_theta") or 0.0
    return c


def tp_slice(cfg, name, rank, tp):
    out = (C.c_int64 * 4)()
    c = make_config(cfg)
    _check(lib().fl_tp_slice(C.byref(c), name.encode(), rank, tp, out))
    return tuple(out)


def _np_dtype_code(a):
    if a.dtype == np.float32:
        return F32
    if a.dtype == np.uint16:
        return BF16
    if a.dtype == np.float16:
        return F16
    raise TypeError("unsupported array dtype %s" % a.dtype)


class Model:
    """fl_model handle.  tensors: dict name -> numpy array (float32 / float16 / uint16 bf16 bits), or
    name -> (device_ptr, dtype_code, shape, device_ordinal) for tensors already in HBM."""

    def __init__(self, cfg, tensors, dtype="bf16", tp_mode=TP_NONE, tp_size=1, tp_rank=0, device_ids=None,
                 unique_id=None):
        L = lib()
        self.cfg = dict(cfg)
        self.V = cfg["vocab_size"]
        arr = (FlTensor * len(tensors))()
        keep = []
        for i, (name, a) in enumerate(tensors.items()):
            arr[i].name = name.encode()
            if isinstance(a, tuple):
                ptr, code, shape, dev = a
                arr[i].dtype, arr[i].ndim, arr[i].data, arr[i].device = code, len(shape), ptr, dev
                for j, s in enumerate(shape):
                    arr[i].shape[j] = s
            else:
                a = np.ascontiguousarray(a)
                keep.append(a)
                arr[i].dtype, arr[i].ndim, arr[i].data, arr[i].device = _np_dtype_code(a), a.ndim, a.ctypes.data, -1
                for j, s in enumerate(a.shape):
                    arr[i].shape[j] = s
        par = FlParallel()
        par.mode, par.tp_size, par.tp_rank = tp_mode, tp_size, tp_rank
        if device_ids is not None:
            ids = (C.c_int32 * len(device_ids))(*device_ids)
            par.device_ids, par.n_device_ids = ids, len(device_ids)
        uid = None
        if unique_id is not None:
            uid = C.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES)
            par.unique_id = C.cast(uid, C.c_void_p)
        c = make_config(cfg)
        h = C.c_void_p()
        _check(L.fl_model_create(C.byref(c), arr, len(tensors), BF16 if dtype == "bf16" else F32, C.byref(par), C.byref(h)))
        self._h = h

    def info(self):
        out = FlModelInfo()
        _check(lib().fl_model_get_info(self._h, C.byref(out)))
        return out

    def new_cache(self, max_seq):
        return Cache(self, max_seq)

    def forward(self, cache, ids, pos):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty(self.V, dtype=np.float32)
        _check(lib().fl_forward(self._h, cache._h, ids.ctypes.data, ids.size, pos, out.ctypes.data))
        return out

    def forward_argmax(self, cache, ids, pos):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        tok = C.c_uint32(0)
        _check(lib().fl_forward_argmax(self._h, cache._h, ids.ctypes.data, ids.size, pos, C.byref(tok)))
        return tok.value

    def decode_greedy(self, cache, first_token, pos, n_steps, eos=-1):
        toks = np.zeros(max(n_steps, 1), dtype=np.uint32)
        n = C.c_size_t(0)
        _check(lib().fl_decode_greedy(self._h, cache._h, int(first_token), pos, n_steps, eos, toks.ctypes.data, C.byref(n)))
        return toks[: n.value]

    def forward_sample(self, cache, ids, pos, temperature, seed=0, draws_done=0):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        tok = C.c_uint32(0)
        sp = FlSampling(temperature, seed, draws_done)
        _check(lib().fl_forward_sample(self._h, cache._h, ids.ctypes.data, ids.size, pos, C.byref(sp), C.byref(tok)))
        return tok.value

    def decode_sample(self, cache, first_token, pos, n_steps, temperature, seed=0, draws_done=1, eos=-1):
        toks = np.zeros(max(n_steps, 1), dtype=np.uint32)
        n = C.c_size_t(0)
        sp = FlSampling(temperature, seed, draws_done)
        _check(lib().fl_decode_sample(self._h, cache._h, int(first_token), pos, n_steps, eos, C.byref(sp), toks.ctypes.data,
                                      C.byref(n)))
        return toks[: n.value]

    def synchronize(self):
        _check(lib().fl_synchronize(self._h))

    def ipc_export(self):
        """This rank's inbox handle (FL_TP_MULTI_PROCESS): ship it to every peer."""
        buf = C.create_string_buffer(IPC_HANDLE_BYTES)
        _check(lib().fl_comm_ipc_export(self._h, buf))
        return buf.raw

    def ipc_connect(self, handles):
        """handles: the tp handles in rank order (own one included)."""
        blob = b"".join(bytes(h) for h in handles)
        buf = C.create_string_buffer(blob, len(blob))
        _check(lib().fl_comm_ipc_connect(self._h, buf))

    def comm_probe(self, form, n, iters=64):
        """us per decode-sized all-reduce on this group's links: form 0 RCCL, 1 one-shot kernel, 2 fused into the GEMV epilogue
        (None: not available).  Collective: every rank calls it."""
        us = C.c_double(-1.0)
        _check(lib().fl_comm_probe(self._h, form, n, iters, C.byref(us)))
        return None if us.value < 0 else us.value

    def comm_selftest(self, n):
        """One all-reduce of n integer-valued floats over the connected inboxes, checked exactly (n >= 16384: the many-workgroup form).
        Collective: every rank calls it.  True: this rank holds the exact sums."""
        ok = C.c_int32(0)
        _check(lib().fl_comm_selftest(self._h, n, C.byref(ok)))
        return bool(ok.value)

    def profile_begin(self):
        _check(lib().fl_profile_begin(self._h))

    def profile_end(self):
        stats = (FlKernelStat * 64)()
        n = C.c_size_t(0)
        _check(lib().fl_profile_end(self._h, stats, 64, C.byref(n)))
        return [dict(name=stats[i].name.decode(), launches=stats[i].launches, total_ms=stats[i].total_ms,
                     bytes=stats[i].bytes, flops=stats[i].flops) for i in range(min(n.value, 64))]

    def close(self):
        if getattr(self, "_h", None):
            lib().fl_model_release(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """fl_batch: up to 64 caches of one model decoded together."""

    def __init__(self, model, caches):
        self._model, self._caches = model, list(caches)
        arr = (C.c_void_p * len(caches))(*[c._h for c in caches])
        h = C.c_void_p()
        _check(lib().fl_batch_create(model._h, arr, len(caches), C.byref(h)))
        self._h = h
        self.n = len(caches)

    def forward(self, tokens, pos, want_logits=True):
        tokens = np.ascontiguousarray(tokens, dtype=np.uint32)
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        lg = np.empty((self.n, self._model.V), dtype=np.float32) if want_logits else None
        am = np.zeros(self.n, dtype=np.uint32)
        _check(lib().fl_batch_forward(self._h, tokens.ctypes.data, pos.ctypes.data, lg.ctypes.data if want_logits else None,
                                      am.ctypes.data))
        return (lg, am) if want_logits else am

    def replace(self, slot, cache):
        """Continuous batching: `cache` takes the place of sequence `slot` (the batch is not rebuilt)."""
        _check(lib().fl_batch_replace(self._h, slot, cache._h))
        self._caches[slot] = cache

    def decode(self, first_tokens, pos, n_steps, eos=-1, temperature=None, seed=0, draws_done=1):
        first = np.ascontiguousarray(first_tokens, dtype=np.uint32)
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.zeros((self.n, max(n_steps, 1)), dtype=np.uint32)
        n_out = np.zeros(self.n, dtype=np.uint64)
        sp = FlSampling(temperature, seed, draws_done) if temperature is not None else None
        _check(lib().fl_batch_decode(self._h, first.ctypes.data, pos.ctypes.data, n_steps, eos, C.byref(sp) if sp else None,
                                     out.ctypes.data, n_out.ctypes.data))
        return [out[i, : int(n_out[i])] for i in range(self.n)]

    def decode_each(self, first_tokens, pos, n_steps, eos=None, temperatures=None, seeds=None, draws_done=None):
        """fl_batch_decode_each: per-sequence EOS ids (None / negative: none) and temperatures (None / < 1e-7: ArgMax)."""
        first = np.ascontiguousarray(first_tokens, dtype=np.uint32)
        pos = np.ascontiguousarray(pos, dtype=np.uint64)
        out = np.zeros((self.n, max(n_steps, 1)), dtype=np.uint32)
        n_out = np.zeros(self.n, dtype=np.uint64)
        e = np.ascontiguousarray([-1 if x is None else int(x) for x in (eos if eos is not None else [None] * self.n)], dtype=np.int64)
        sp = (FlSampling * self.n)()
        for i in range(self.n):
            t = temperatures[i] if temperatures is not None and temperatures[i] is not None else 0.0
            sp[i] = FlSampling(t, seeds[i] if seeds is not None else 0, draws_done[i] if draws_done is not None else 1)
        _check(lib().fl_batch_decode_each(self._h, first.ctypes.data, pos.ctypes.data, n_steps, e.ctypes.data, C.cast(sp, C.c_void_p),
                                          out.ctypes.data, n_out.ctypes.data))
        return [out[i, : int(n_out[i])] for i in range(self.n)]

    def close(self):
        if getattr(self, "_h", None):
            lib().fl_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Cache:
    def __init__(self, model, max_seq):
        h = C.c_void_p()
        _check(lib().fl_cache_create(model._h, max_seq, C.byref(h)))
        self._h = h
        self._model = model

    def reset(self):
        lib().fl_cache_reset(self._h)

    def __len__(self):
        return lib().fl_cache_len(self._h)

    def capacity(self):
        return lib().fl_cache_capacity(self._h)

    def close(self):
        if getattr(self, "_h", None):
            lib().fl_cache_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tune(key, value):
    _check(lib().fl_tune(key.encode(), int(value)))


def reload_env():
    """The library reads its FL_<NAME> switches from the environment ONCE (first use); after changing os.environ in a live process
    call this to have every switch re-read."""
    _check(lib().fl_tune(b"reload_env", 0))


def library_loaded():
    return _LIB is not None


def op_attention(q, k, v, s_past, H, Hkv, d, window=-1, kernel=0, nsplit=0):
    """The bf16 MFMA attention kernels alone.  q [T, H*d], k / v [s_past + T, Hkv*d]: uint16 bf16 bits; returns [T, H*d] f32."""
    q, k, v = (np.ascontiguousarray(a, dtype=np.uint16) for a in (q, k, v))
    T = q.shape[0]
    assert q.shape == (T, H * d) and k.shape == (s_past + T, Hkv * d) and v.shape == k.shape
    out = np.empty((T, H * d), dtype=np.float32)
    _check(lib().fl_op_attention(q.ctypes.data, k.ctypes.data, v.ctypes.data, T, s_past, H, Hkv, d, window, kernel, nsplit, out.ctypes.data))
    return out


def op_linear(x, w, bias=None, epilogue=0, iters=0):
    """y = x . w^T (+bias) through the projection kernels.  x [T,K], w [N,K]: both float32 or both uint16 (bf16)."""
    x = np.ascontiguousarray(x)
    w = np.ascontiguousarray(w)
    assert x.dtype == w.dtype
    T, K = x.shape
    N = w.shape[0]
    y = np.empty((T, N // 2 if epilogue == 1 else N), dtype=np.float32)
    b = np.ascontiguousarray(bias, dtype=np.float32) if bias is not None else None
    ms = C.c_double(0.0)
    _check(lib().fl_op_linear(x.ctypes.data, w.ctypes.data, b.ctypes.data if b is not None else None, T, N, K,
                              _np_dtype_code(x), epilogue, y.ctypes.data, iters, C.byref(ms)))
    return (y, ms.value) if iters else y
