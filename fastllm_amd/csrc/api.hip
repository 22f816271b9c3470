// api.hip -- the extern "C" surface declared in include/fastllm_mi355x.h.
//
// Every entry point is an exception barrier: the host of this library is Rust (or ctypes), and a C++
// exception that crosses the C ABI is undefined behaviour there.  The reference surfaces failures as
// anyhow::Error (mod.rs:402-405,446-451); here std::bad_alloc becomes FL_ERR_OOM and anything else
// FL_ERR_HIP, both with fl_last_error() set.
#include <algorithm>
#include <exception>
#include <memory>
#include <new>
#include <vector>

#include "model.h"

using namespace fl;

// fl_model / fl_cache stay opaque: the handles are the C++ objects themselves

static Model *M(fl_model *m) { return reinterpret_cast<Model *>(m); }
static const Model *M(const fl_model *m) { return reinterpret_cast<const Model *>(m); }
static Cache *C(fl_cache *c) { return reinterpret_cast<Cache *>(c); }
static const Cache *C(const fl_cache *c) { return reinterpret_cast<const Cache *>(c); }

template <typename F> static int guarded(F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        set_error("out of host memory (std::bad_alloc)");
        return FL_ERR_OOM;
    } catch (const std::exception &e) {
        set_error("internal error: %s", e.what());
        return FL_ERR_HIP;
    } catch (...) {
        set_error("internal error: unknown C++ exception");
        return FL_ERR_HIP;
    }
}
template <typename F> static void guarded_void(F &&body) noexcept {
    (void)guarded([&]() -> int { body(); return FL_OK; });
}

extern "C" {

int fl_abi_version(void) { return FL_ABI_VERSION; }
const char *fl_last_error(void) { return last_error(); }

int fl_device_count(int *count) {
    return guarded([&]() -> int {
        if (!count) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null count");
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
        *count = n;
        return FL_OK;
    });
}

int fl_comm_unique_id(void *out) {
    return guarded([&]() -> int {
        if (!out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null out");
        static_assert(sizeof(ncclUniqueId) <= FL_UNIQUE_ID_BYTES, "unique id size");
        ncclUniqueId id;
        ncclResult_t r = ncclGetUniqueId(&id);
        if (r != ncclSuccess) FL_FAIL(FL_ERR_RCCL, "ncclGetUniqueId: %s", ncclGetErrorString(r));
        memset(out, 0, FL_UNIQUE_ID_BYTES);
        memcpy(out, &id, sizeof id);
        return FL_OK;
    });
}

int fl_comm_ipc_export(fl_model *m, void *handle_out) {
    return guarded([&]() -> int {
        return comm_ipc_export(M(m), handle_out);
    });
}
int fl_comm_ipc_connect(fl_model *m, const void *handles) {
    return guarded([&]() -> int {
        return comm_ipc_connect(M(m), handles);
    });
}

int fl_model_create(const fl_config *cfg, const fl_tensor *tensors, size_t n_tensors, int32_t compute_dtype,
                    const fl_parallel *par, fl_model **out) {
    return guarded([&]() -> int {
        if (!out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "fl_model_create: null out");
        *out = nullptr;
        Model *m = nullptr;
        int rc = model_create(cfg, tensors, n_tensors, compute_dtype, par, &m);
        if (rc == FL_OK) *out = reinterpret_cast<fl_model *>(m);
        return rc;
    });
}

void fl_model_retain(fl_model *m) {
    guarded_void([&]() {
        if (m) M(m)->refs.fetch_add(1);
    });
}
void fl_model_release(fl_model *m) {
    guarded_void([&]() {
        if (m && M(m)->refs.fetch_sub(1) == 1) delete M(m);
    });
}

int fl_model_get_info(const fl_model *m, fl_model_info *out) {
    return guarded([&]() -> int {
        if (!m || !out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
        const Model *mm = M(m);
        const Dims &D = mm->D;
        memset(out, 0, sizeof *out);
        out->cfg = mm->cfg_resolved;
        out->head_dim = D.dm;                                    // the model's (the kernels may run it padded to 64 / 128)
        out->compute_dtype = mm->dtype;
        out->tp_size = mm->tp;
        const int64_t es = (int64_t)mm->esize();
        // weights read once per decoded token (one embedding row, norms, biases
        // included; the rest of the embedding table excluded)
        int64_t per_layer = 2 * D.h * D.h + 2 * D.Hkv * D.dm * D.h + 3 * D.h * D.inter;
        int64_t small = 2 * D.h + (D.qkv_bias ? D.h + 2 * D.Hkv * D.dm : 0);
        out->weight_bytes_per_token = es * (D.L * (per_layer + small) + D.h + D.V * D.h);
        out->kv_bytes_per_position = es * D.L * D.Hkv * D.dm * 2;
        out->hbm_bytes_allocated = mm->hbm_bytes;
        // + the GEMM workspaces of each shard's streams (first long prompt).  The workspace tables are keyed by (current device,
        // stream): the calling thread's device is put back afterwards, and forward() is kept out meanwhile (it sets devices too).
        std::lock_guard<std::mutex> lock(const_cast<Model *>(mm)->mu);
        int dev0 = -1;
        (void)hipGetDevice(&dev0);
        struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{dev0};
        for (auto &sh : mm->shards) {
            if (hipSetDevice(sh.device) != hipSuccess) continue;
            out->hbm_bytes_allocated += gemm_8p_workspace_bytes(sh.stream) + gemm_8p_workspace_bytes(sh.comm_stream) + gemm_h4_workspace_bytes(sh.stream) +
                                        gemm_h4_workspace_bytes(sh.comm_stream);
        }
        out->small_collectives = mm->shards[0].pc.connected ? 2 : mm->tp == 1 ? 0 : mm->tp_mode == FL_TP_EMULATED ? 3 : 1;
        out->fused_all_reduce = fused_all_reduce_ready(mm) ? 1 : 0;
        if (mm->shards[0].comm) {
            int n = 0;
            if (ncclCommCount(mm->shards[0].comm, &n) == ncclSuccess) out->rccl_ranks = n;
        }
        return FL_OK;
    });
}

int fl_cache_create(fl_model *m, size_t max_seq, fl_cache **out) {
    return guarded([&]() -> int {
        if (!out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "fl_cache_create: null out");
        *out = nullptr;
        Cache *c = nullptr;
        int rc = cache_create(M(m), max_seq, &c);
        if (rc == FL_OK) *out = reinterpret_cast<fl_cache *>(c);
        return rc;
    });
}
void fl_cache_reset(fl_cache *c) {
    guarded_void([&]() {
        if (c) C(c)->len = 0;
    });
}
size_t fl_cache_len(const fl_cache *c) { return c ? C(c)->len : 0; }
size_t fl_cache_capacity(const fl_cache *c) { return c ? C(c)->max_seq : 0; }
void fl_cache_destroy(fl_cache *c) {
    guarded_void([&]() {
        if (!c) return;
        Model *m = C(c)->m;
        delete C(c);
        if (m && m->refs.fetch_sub(1) == 1) delete m;
    });
}

int fl_forward(fl_model *m, fl_cache *c, const uint32_t *ids, size_t T, size_t pos, float *logits_out) {
    return guarded([&]() -> int {
        if (!logits_out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null logits_out");
        return forward(M(m), C(c), ids, T, pos, logits_out, nullptr);
    });
}

int fl_forward_argmax(fl_model *m, fl_cache *c, const uint32_t *ids, size_t T, size_t pos, uint32_t *token_out) {
    return guarded([&]() -> int {
        if (!token_out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null token_out");
        return forward(M(m), C(c), ids, T, pos, nullptr, token_out);
    });
}

int fl_decode_greedy(fl_model *m, fl_cache *c, uint32_t first_token, size_t pos, size_t n_steps, int64_t eos,
                     uint32_t *tokens_out, size_t *n_out) {
    return guarded([&]() -> int {
        return decode_greedy(M(m), C(c), first_token, pos, n_steps, eos, tokens_out, n_out);
    });
}

int fl_forward_sample(fl_model *m, fl_cache *c, const uint32_t *ids, size_t T, size_t pos, const fl_sampling *sampling,
                      uint32_t *token_out) {
    return guarded([&]() -> int {
        if (!token_out || !sampling) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
        return forward(M(m), C(c), ids, T, pos, nullptr, token_out, sampling);
    });
}

int fl_decode_sample(fl_model *m, fl_cache *c, uint32_t first_token, size_t pos, size_t n_steps, int64_t eos,
                     const fl_sampling *sampling, uint32_t *tokens_out, size_t *n_out) {
    return guarded([&]() -> int {
        if (!sampling) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null sampling");
        return decode_greedy(M(m), C(c), first_token, pos, n_steps, eos, tokens_out, n_out, sampling);
    });
}

int fl_batch_create(fl_model *m, fl_cache *const *caches, size_t n, fl_batch **out) {
    return guarded([&]() -> int {
        if (!out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null out");
        Batch *b = nullptr;
        FL_TRY(batch_create(M(m), reinterpret_cast<Cache *const *>(caches), n, &b));
        *out = reinterpret_cast<fl_batch *>(b);
        return FL_OK;
    });
}
void fl_batch_destroy(fl_batch *b) {
    guarded_void([&]() {
        if (!b) return;
        Batch *bb = reinterpret_cast<Batch *>(b);
        Model *m = bb->m;
        delete bb;
        if (m && m->refs.fetch_sub(1) == 1) delete m;
    });
}
int fl_batch_replace(fl_batch *b, size_t slot, fl_cache *cache) {
    return guarded([&]() -> int {
        return batch_replace(reinterpret_cast<Batch *>(b), slot, reinterpret_cast<Cache *>(cache));
    });
}
int fl_batch_forward(fl_batch *b, const uint32_t *tokens, const size_t *pos, float *logits_out, uint32_t *argmax_out) {
    return guarded([&]() -> int {
        return batch_forward(reinterpret_cast<Batch *>(b), tokens, pos, logits_out, argmax_out);
    });
}
int fl_batch_decode(fl_batch *b, const uint32_t *first_tokens, const size_t *pos, size_t n_steps, int64_t eos,
                    const fl_sampling *sampling, uint32_t *tokens_out, size_t *n_out) {
    return guarded([&]() -> int {
        return batch_decode(reinterpret_cast<Batch *>(b), first_tokens, pos, n_steps, eos, sampling, tokens_out, n_out);
    });
}
int fl_batch_decode_each(fl_batch *b, const uint32_t *first_tokens, const size_t *pos, size_t n_steps, const int64_t *eos,
                         const fl_sampling *sampling, uint32_t *tokens_out, size_t *n_out) {
    return guarded([&]() -> int {
        return batch_decode_each(reinterpret_cast<Batch *>(b), first_tokens, pos, n_steps, eos, sampling, tokens_out, n_out);
    });
}

int fl_synchronize(fl_model *m) {
    return guarded([&]() -> int {
        if (!m) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null model");
        for (auto &sh : M(m)->shards) { FL_HIP(hipSetDevice(sh.device)); FL_HIP(hipStreamSynchronize(sh.stream)); }
        return FL_OK;
    });
}

int fl_comm_probe(fl_model *m, int32_t form, int64_t n, int32_t iters, double *us_per_call) {
    return guarded([&]() -> int {
        if (!m) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null model");
        return comm_probe(M(m), form, n, iters, us_per_call);
    });
}

int fl_comm_selftest(fl_model *m, int64_t n, int32_t *ok) {
    return guarded([&]() -> int {
        if (!m || !ok) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
        int good = 0;
        const int rc = comm_selftest(M(m), n, &good);
        *ok = good;
        return rc;
    });
}

int fl_tp_slice(const fl_config *cfg, const char *tensor_name, int32_t tp_rank, int32_t tp_size, int64_t out[4]) {
    return guarded([&]() -> int {
        if (!tensor_name || !out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
        Dims D;
        FL_TRY(resolve_config(cfg, &D));
        return tp_slice(D, tensor_name, tp_rank, tp_size, out);
    });
}

int fl_profile_begin(fl_model *m) {
    return guarded([&]() -> int {
        if (!m) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null model");
        std::lock_guard<std::mutex> lock(M(m)->mu);
        for (auto &r : M(m)->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        M(m)->prof.clear();
        M(m)->profiling = true;
        return FL_OK;
    });
}

int fl_profile_end(fl_model *m, fl_kernel_stat *stats, size_t cap, size_t *n_stats) {
    return guarded([&]() -> int {
        if (!m || !n_stats) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
        Model *mm = M(m);
        std::lock_guard<std::mutex> lock(mm->mu);
        mm->profiling = false;
        for (auto &sh : mm->shards) { FL_HIP(hipSetDevice(sh.device)); FL_HIP(hipStreamSynchronize(sh.stream)); }
        std::vector<fl_kernel_stat> acc;
        for (auto &r : mm->prof) {
            float ms = 0.f;
            FL_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
            char name[48];
            if (r.tag[0]) snprintf(name, sizeof name, "%s[%s]", kernel_class_name(r.kc), r.tag);
            else snprintf(name, sizeof name, "%s", kernel_class_name(r.kc));
            fl_kernel_stat *st = nullptr;
            for (auto &e : acc) if (!strcmp(e.name, name)) { st = &e; break; }
            if (!st) { fl_kernel_stat e; memset(&e, 0, sizeof e); snprintf(e.name, sizeof e.name, "%s", name); acc.push_back(e); st = &acc.back(); }
            st->launches++; st->total_ms += ms; st->bytes += r.bytes; st->flops += r.flops;
        }
        for (auto &r : mm->prof) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
        mm->prof.clear();
        size_t n = 0;
        for (auto &e : acc) {
            if (stats && n < cap) stats[n] = e;
            n++;
        }
        *n_stats = n;
        return FL_OK;
    });
}

int fl_tune(const char *key, int value) {
    return guarded([&]() -> int {
        if (!key || value < -1) FL_FAIL(FL_ERR_BAD_ARGUMENT, "bad tuning key/value");       // (-1: "automatic", for the switches that have it)
        if (!strcmp(key, "gemv_blocks")) gemv_set_tuning(value, -1);                    // 0 = automatic
        else if (!strcmp(key, "gemv_waves")) gemv_set_tuning(-1, value);                // 0 = automatic
        else if (!strcmp(key, "engine_grid")) {                                         // 0 = one workgroup per CU (tests: a grid that cannot be resident)
#ifndef FL_EXPERIMENTAL
            FL_FAIL(FL_ERR_UNSUPPORTED, "default build: engine_grid belongs to the experimental decode engine");
#endif
            engine_set_grid(value);
        }
        else if (!strcmp(key, "experimental")) {                                        // is this the EXPERIMENTAL build? (tests skip otherwise)
#ifndef FL_EXPERIMENTAL
            FL_FAIL(FL_ERR_UNSUPPORTED, "default build: the experimental kernels (decode engine, fused attention + o_proj, attention prefetch, loader waves) are not compiled in");
#endif
        }
        else if (!strcmp(key, "reload_env")) tune_reload_env();                        // re-read every FL_<NAME> switch of the table (common.h)
        else {
            const int rc = tune_set(key, value);
            if (rc == FL_ERR_UNSUPPORTED) FL_FAIL(FL_ERR_UNSUPPORTED, "default build: %s is a switch of a kernel compiled into the EXPERIMENTAL build only", key);
            if (rc != FL_OK) FL_FAIL(FL_ERR_BAD_ARGUMENT, "unknown tuning key %s", key);
        }
        return FL_OK;
    });
}

int fl_op_sample(const float *logits, int64_t V, const fl_sampling *sampling, int64_t n_draws, uint32_t *tokens_out) {
    return guarded([&]() -> int {
        if (!logits || !sampling || !tokens_out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
        if (V <= 0 || V > (1 << 24) || n_draws <= 0 || n_draws > (1 << 20)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "bad size");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) FL_FAIL(FL_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU path");
        FL_HIP(hipSetDevice(0));
        struct Bufs { float *lg = 0, *sc = 0; StepState *st = 0; SampleState *ss = 0; uint32_t *out = 0; hipStream_t s = 0;
                      ~Bufs() { (void)hipFree(lg); (void)hipFree(sc); (void)hipFree(st); (void)hipFree(ss); (void)hipFree(out); if (s) (void)hipStreamDestroy(s); } } B;
        FL_HIP(hipStreamCreate(&B.s));
        FL_HIP(hipMalloc((void **)&B.lg, (size_t)V * 4));
        FL_HIP(hipMalloc((void **)&B.sc, (size_t)V * 4));
        FL_HIP(hipMalloc((void **)&B.st, sizeof(StepState)));
        FL_HIP(hipMalloc((void **)&B.ss, sizeof(SampleState)));
        FL_HIP(hipMalloc((void **)&B.out, (size_t)n_draws * 4));
        StepState st{}; st.eos = -1;
        const SampleState ss = make_sampler(sampling);
        FL_HIP(hipMemcpy(B.lg, logits, (size_t)V * 4, hipMemcpyHostToDevice));
        FL_HIP(hipMemcpy(B.st, &st, sizeof st, hipMemcpyHostToDevice));
        FL_HIP(hipMemcpy(B.ss, &ss, sizeof ss, hipMemcpyHostToDevice));
        Launcher L; L.stream = B.s;
        for (int64_t i = 0; i < n_draws; i++) FL_TRY(launch_select_advance(L, B.lg, V, B.st, B.ss, B.sc, B.out, 1));
        FL_HIP(hipStreamSynchronize(B.s));
        FL_HIP(hipMemcpy(tokens_out, B.out, (size_t)n_draws * 4, hipMemcpyDeviceToHost));
        return FL_OK;
    });
}

int fl_op_linear(const void *x, const void *w, const float *bias, int64_t T, int64_t N, int64_t K, int32_t dtype,
                 int32_t epilogue, float *y, int32_t iters, double *ms_out) {
    return guarded([&]() -> int {
        if (!x || !w || !y) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
        if (dtype != FL_DTYPE_BF16 && dtype != FL_DTYPE_F32) FL_FAIL(FL_ERR_UNSUPPORTED, "dtype must be bf16 or f32");
        if (T <= 0 || N <= 0 || K <= 0 || K % 8) FL_FAIL(FL_ERR_BAD_ARGUMENT, "bad shape (K must be a multiple of 8)");
        if (epilogue == EPI_GATEUP && (N % 2)) FL_FAIL(FL_ERR_BAD_ARGUMENT, "gate/up needs an even row count");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) FL_FAIL(FL_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU path");
        FL_HIP(hipSetDevice(0));
        const size_t es = dtype == FL_DTYPE_BF16 ? 2 : 4;
        const int64_t I = N / 2, Ip = (I + 15) / 16 * 16;
        const int64_t Nw = epilogue == EPI_GATEUP ? 2 * Ip : N;          // rows of the device matrix
        const int64_t Ny = epilogue == EPI_GATEUP ? Ip : N;              // columns of the device output
        struct Bufs { void *x = 0, *w = 0, *ws = 0, *y = 0; float *b = 0; hipStream_t s = 0; hipEvent_t e0 = 0, e1 = 0;
                      std::vector<void *> copies;
                      ~Bufs() { (void)hipFree(x); (void)hipFree(w); (void)hipFree(ws); (void)hipFree(y); (void)hipFree(b);
                                for (size_t i = 1; i < copies.size(); i++) (void)hipFree(copies[i]);
                                if (s) { (void)hipStreamSynchronize(s); gemm_8p_release_stream(s); gemm_h4_release_stream(s); (void)hipStreamDestroy(s); }
                                if (e0) (void)hipEventDestroy(e0);
                                if (e1) (void)hipEventDestroy(e1); } } B;
        FL_HIP(hipStreamCreate(&B.s));
        FL_HIP(hipMalloc(&B.x, (size_t)T * K * es));
        FL_HIP(hipMalloc(&B.w, (size_t)Nw * K * es));
        const size_t ybytes = (size_t)T * Ny * (epilogue == EPI_GATEUP ? es : 4);
        const int op_split = tune(TK_OP_MAXSPLIT);
        const int max_split = (epilogue == EPI_F32 && !bias) ? (op_split > 0 ? op_split : std::max(4, ksplit_cap(T))) : 1;      // exercise split-K where the model would
        int nsplit = 1;
        FL_HIP(hipMalloc(&B.y, ybytes * max_split));
        FL_HIP(hipMemcpy(B.x, x, (size_t)T * K * es, hipMemcpyHostToDevice));
        Launcher L; L.stream = B.s;
        if (epilogue == EPI_GATEUP) {
            FL_HIP(hipMalloc(&B.ws, (size_t)N * K * es));
            FL_HIP(hipMemcpy(B.ws, w, (size_t)N * K * es, hipMemcpyHostToDevice));
            FL_HIP(hipMemsetAsync(B.w, 0, (size_t)Nw * K * es, B.s));
            FL_TRY(launch_convert_slice(L, dtype, B.ws, K, 0, 0, I, K, dtype, B.w, K, 0, 1));
            FL_TRY(launch_convert_slice(L, dtype, B.ws, K, I, 0, I, K, dtype, B.w, K, 0, 2));
        } else {
            FL_HIP(hipMemcpy(B.w, w, (size_t)N * K * es, hipMemcpyHostToDevice));
            if (bias) {
                FL_HIP(hipMalloc((void **)&B.b, (size_t)N * 4));
                FL_HIP(hipMemcpy(B.b, bias, (size_t)N * 4, hipMemcpyHostToDevice));
            }
        }
        // FL_OP_LINEAR_DMA=1: T <= 8 rows go through the batched-decode projection kernel (k_gemv_dma.hip) instead, so that
        // its weight-streaming rate can be measured (and its arithmetic tested) without a model around it
        const bool use_dma = tune(TK_OP_LINEAR_DMA) == 1;
        const bool dma = use_dma && dtype == FL_DTYPE_BF16 && T <= 32 && !B.b && gemv_dma_supported((int)T, Nw, K, epilogue, 0) &&
                         gemv_dma_ksplit(K, Nw, epilogue) <= max_split;
        auto run = [&](const void *wp) -> int {
            if (!dma) return launch_linear(L, dtype, wp, B.x, B.b, B.y, T, Nw, K, epilogue, nullptr, max_split, &nsplit);
            GemvBatchArgs ga;
            ga.W = wp; ga.x = B.x; ga.out = B.y; ga.N = (int)Nw; ga.K = (int)K; ga.epi = epilogue; ga.pro = PRO_X; ga.B = (int)T;
            ga.nks = epilogue == EPI_F32 ? gemv_dma_ksplit(K, Nw, epilogue) : 1;
            nsplit = ga.nks;
            return launch_gemv_dma(L, ga);
        };
        FL_TRY(run(B.w));
        FL_HIP(hipStreamSynchronize(B.s));
        if (iters > 0 && ms_out) {
            // the timed launches rotate over copies of W that together exceed the 256 MiB Infinity Cache: in the forward pass a
            // projection's weights always come from HBM, and a back-to-back replay on ONE copy would read them from the cache
            const size_t wbytes = (size_t)Nw * K * es;
            const int hot = tune(TK_OP_HOT);   // 1: one copy (L2 + Infinity Cache); n > 1: n copies (past the L2s, inside the Infinity Cache when n x bytes < 256 MiB)
            const int ncopy = hot > 0 ? hot : (int)std::min<size_t>(24, std::max<size_t>(1, (640u << 20) / wbytes + 1));
            std::vector<void *> &copies = B.copies;
            copies.push_back(nullptr);                                   // slot 0 = B.w itself
            for (int c = 1; c < ncopy; c++) {
                void *p = nullptr;
                FL_HIP(hipMalloc(&p, wbytes));
                copies.push_back(p);
                FL_HIP(hipMemcpyAsync(p, B.w, wbytes, hipMemcpyDeviceToDevice, B.s));
            }
            copies[0] = B.w;
            for (int c = 0; c < ncopy; c++) FL_TRY(run(copies[c]));   // warm
            FL_HIP(hipStreamSynchronize(B.s));
            FL_HIP(hipEventCreate(&B.e0)); FL_HIP(hipEventCreate(&B.e1));
            FL_HIP(hipEventRecord(B.e0, B.s));
            for (int i = 0; i < iters; i++) FL_TRY(run(copies[i % ncopy]));
            FL_HIP(hipEventRecord(B.e1, B.s));
            FL_HIP(hipEventSynchronize(B.e1));
            float ms = 0.f; FL_HIP(hipEventElapsedTime(&ms, B.e0, B.e1));
            *ms_out = ms / iters;
        }
        if (epilogue == EPI_GATEUP) {
            std::unique_ptr<unsigned char[]> tmp(new unsigned char[ybytes]);
            FL_HIP(hipMemcpy(tmp.get(), B.y, ybytes, hipMemcpyDeviceToHost));
            for (int64_t t = 0; t < T; t++)
                for (int64_t j = 0; j < I; j++) {
                    const size_t idx = (size_t)t * Ip + j;
                    y[(size_t)t * I + j] = es == 2 ? bf16_bits_to_float(reinterpret_cast<bf16_t *>(tmp.get())[idx])
                                                   : reinterpret_cast<float *>(tmp.get())[idx];
                }
        } else {
            FL_HIP(hipMemcpy(y, B.y, ybytes, hipMemcpyDeviceToHost));
            if (nsplit > 1) {                                   // sum the split-K slabs in slab order, like rmsnorm_add does
                std::unique_ptr<float[]> tmp(new float[(size_t)T * Ny]);
                for (int sl = 1; sl < nsplit; sl++) {
                    FL_HIP(hipMemcpy(tmp.get(), (char *)B.y + (size_t)sl * ybytes, ybytes, hipMemcpyDeviceToHost));
                    for (size_t i = 0; i < (size_t)T * Ny; i++) y[i] += tmp[i];
                }
            }
        }
        return FL_OK;
    });
}

int fl_op_attention(const void *q, const void *k, const void *v, int64_t T, int64_t s_past, int64_t H, int64_t Hkv, int64_t d,
                    int64_t window, int32_t kernel, int32_t nsplit, float *out) {
    return guarded([&]() -> int {
        if (!q || !k || !v || !out) FL_FAIL(FL_ERR_BAD_ARGUMENT, "null argument");
        if (T < 1 || s_past < 0 || H < 1 || Hkv < 1 || kernel < 0 || kernel > 3 || nsplit < 0 || nsplit > 64) FL_FAIL(FL_ERR_BAD_ARGUMENT, "bad shape / kernel");
        if (!attn_mfma_supported(FL_DTYPE_BF16, H, Hkv, d)) FL_FAIL(FL_ERR_UNSUPPORTED, "MFMA attention: head_dim 64 / 128, at most 8 query heads per kv head");
        if (kernel == 1 && T != 1) FL_FAIL(FL_ERR_BAD_ARGUMENT, "the decode kernel takes one query token");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) FL_FAIL(FL_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU path");
        FL_HIP(hipSetDevice(0));
        const int64_t S = s_past + T, sa = (S + 31) / 32 * 32;
        // the cache layout of the model: K [Hkv][sa][d], V transposed [Hkv][d][sa], zero padding (model.hip cache_create)
        std::vector<bf16_t> kc((size_t)Hkv * sa * d, 0), vt((size_t)Hkv * d * sa, 0);
        const bf16_t *kh = reinterpret_cast<const bf16_t *>(k), *vh = reinterpret_cast<const bf16_t *>(v);
        for (int64_t s = 0; s < S; s++)
            for (int64_t h = 0; h < Hkv; h++)
                for (int64_t j = 0; j < d; j++) {
                    kc[((size_t)h * sa + s) * d + j] = kh[((size_t)s * Hkv + h) * d + j];
                    vt[((size_t)h * d + j) * sa + s] = vh[((size_t)s * Hkv + h) * d + j];
                }
        if (nsplit == 0) nsplit = (int)std::max<int64_t>(1, std::min<int64_t>((S + 127) / 128, 48));
        struct Bufs { void *q = 0, *k = 0, *v = 0, *o = 0; StepState *st = 0; float *pm = 0, *pl = 0, *po = 0; unsigned *cnt = 0; hipStream_t s = 0;
                      ~Bufs() { (void)hipFree(q); (void)hipFree(k); (void)hipFree(v); (void)hipFree(o); (void)hipFree(st); (void)hipFree(pm);
                                (void)hipFree(pl); (void)hipFree(po); (void)hipFree(cnt); if (s) (void)hipStreamDestroy(s); } } B;
        const size_t qb = (size_t)T * H * d * 2;
        FL_HIP(hipStreamCreate(&B.s));
        FL_HIP(hipMalloc(&B.q, qb)); FL_HIP(hipMalloc(&B.o, qb));
        FL_HIP(hipMalloc(&B.k, kc.size() * 2)); FL_HIP(hipMalloc(&B.v, vt.size() * 2));
        FL_HIP(hipMalloc((void **)&B.st, sizeof(StepState)));
        FL_HIP(hipMalloc((void **)&B.pm, (size_t)H * nsplit * 4)); FL_HIP(hipMalloc((void **)&B.pl, (size_t)H * nsplit * 4));
        FL_HIP(hipMalloc((void **)&B.po, (size_t)H * nsplit * d * 4)); FL_HIP(hipMalloc((void **)&B.cnt, (size_t)H * 4));
        StepState st{}; st.pos = (uint32_t)s_past; st.len = (uint32_t)s_past; st.call0 = (uint32_t)s_past; st.eos = -1;
        FL_HIP(hipMemcpy(B.q, q, qb, hipMemcpyHostToDevice));
        FL_HIP(hipMemcpy(B.k, kc.data(), kc.size() * 2, hipMemcpyHostToDevice));
        FL_HIP(hipMemcpy(B.v, vt.data(), vt.size() * 2, hipMemcpyHostToDevice));
        FL_HIP(hipMemcpy(B.st, &st, sizeof st, hipMemcpyHostToDevice));
        FL_HIP(hipMemset(B.cnt, 0, (size_t)H * 4));
        FL_HIP(hipMemset(B.o, 0xff, qb));                         // NaN pattern: an element the kernel does not write shows up
        Launcher L; L.stream = B.s;
        const float scale = 1.0f / sqrtf((float)d);
        int rc;
        if (kernel == 1 || (kernel == 0 && T == 1)) {
            AttnScratch as{B.pm, B.pl, B.po, B.cnt, nsplit, S};
            rc = launch_attn_decode_mfma(L, B.q, B.k, B.v, B.st, B.o, as, H, Hkv, d, sa, scale);
        } else {
            attn_prefill_force(kernel);
            rc = launch_attn_prefill_mfma(L, B.q, B.k, B.v, B.st, B.o, T, H, Hkv, d, sa, scale, window < 0 ? -1 : window);
            attn_prefill_force(0);
        }
        FL_TRY(rc);
        FL_HIP(hipStreamSynchronize(B.s));
        std::vector<bf16_t> oh((size_t)T * H * d);
        FL_HIP(hipMemcpy(oh.data(), B.o, qb, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < oh.size(); i++) out[i] = bf16_bits_to_float(oh[i]);
        return FL_OK;
    });
}

}  // extern "C"
