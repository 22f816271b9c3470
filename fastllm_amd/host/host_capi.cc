// host_capi.cc -- a small C surface over fastllm_host.hpp so that pytest (ctypes) can drive the C++
// mirror of the reference's trait layer.  Built into fastllm_amd/lib/libfastllm_host.so.
#include <cstdio>
#include <string>

#include "fastllm_host.hpp"
#include "safetensors.hpp"

using namespace fastllm;

namespace {
thread_local std::string g_err;

struct Handle {
    int family;                               // 0 llama, 1 mistral, 2 qwen
    std::unique_ptr<Model<LlamaWithConfig>> llama;
    std::unique_ptr<Model<MistralWithConfig>> mistral;
    std::unique_ptr<Model<QwenWithConfig>> qwen;
};

template <class F> int guard(F &&f) {
    try { return f(); }
    catch (const Error &e) { g_err = e.what(); return e.code ? e.code : -100; }
    catch (const Panic &e) { g_err = std::string("panic: ") + e.what(); return -101; }
    catch (const std::exception &e) { g_err = e.what(); return -102; }
}

TensorMap to_map(const fl_tensor *ts, size_t n) {
    TensorMap m;
    for (size_t i = 0; i < n; i++) {
        Tensor t; t.dtype = (DType)ts[i].dtype; t.data = ts[i].data; t.device = ts[i].device;
        for (int d = 0; d < ts[i].ndim; d++) t.shape.push_back(ts[i].shape[d]);
        m[ts[i].name] = t;
    }
    return m;
}
}  // namespace

extern "C" {

const char *flh_last_error(void) { return g_err.c_str(); }

const char *flh_get_family(int family) {
    switch (family) { case 0: return LlamaWithConfig::get_family(); case 1: return MistralWithConfig::get_family(); case 2: return QwenWithConfig::get_family(); }
    return "";
}

int flh_supports_architecture(int family, const char *arch) {
    switch (family) {
        case 0: return LlamaWithConfig::supports_architecture(arch);
        case 1: return MistralWithConfig::supports_architecture(arch);
        case 2: return QwenWithConfig::supports_architecture(arch);
    }
    return 0;
}

// parse config.json and run the family's validation; out (optional) receives the resolved fl_config
int flh_config_check(int family, const char *json, fl_config *out) {
    return guard([&] {
        BaseModelConfig c = BaseModelConfig::from_json(json);
        if (family == 1) MistralWithConfig::validate(c);
        if (family == 2) {
            try { c.validate_head_dimensions(); c.validate_gqa_config(); }
            catch (const Error &e) { throw Panic(e.what()); }
        }
        if (out) *out = c.to_fl((fl_family)family, family == 2);
        return 0;
    });
}

int flh_model_create(int family, const char *config_json, const fl_tensor *tensors, size_t n, int dtype,
                     int device_ordinal, void **out) {
    return guard([&] {
        BaseModelConfig cfg = BaseModelConfig::from_json(config_json);
        TensorMap map = to_map(tensors, n);
        Device dev = device_ordinal < 0 ? Device::cpu() : Device::mi355x(device_ordinal);
        auto h = std::make_unique<Handle>();
        h->family = family;
        if (family == 0) { auto r = LlamaWithConfig::initialize_model(cfg, map, (DType)dtype, dev); h->llama = std::make_unique<Model<LlamaWithConfig>>(r.first, dev, r.second); }
        else if (family == 1) { auto r = MistralWithConfig::initialize_model(cfg, map, (DType)dtype, dev); h->mistral = std::make_unique<Model<MistralWithConfig>>(r.first, dev, r.second); }
        else if (family == 2) { auto r = QwenWithConfig::initialize_model(cfg, map, (DType)dtype, dev); h->qwen = std::make_unique<Model<QwenWithConfig>>(r.first, dev, r.second); }
        else throw Error(FL_ERR_BAD_ARGUMENT, "unknown family");
        *out = h.release();
        return 0;
    });
}

// load_model::<M>(dir) (huggingface.rs:18-139 minus Hub download and tokenizer): the family is chosen from
// config.json's architectures[0] the way the registry does (model_registry.rs:169-182)
int flh_load_dir(const char *dir, int dtype, int device_ordinal, void **out, int *family_out) {
    return guard([&] {
        const std::string d(dir);
        const std::string arch = architecture_of(read_text(d + "/config.json"));
        const char *fam = get_family_from_architecture(arch);
        if (!fam) throw Error(FL_ERR_BAD_CONFIG, "Unsupported architecture " + arch);
        Device dev = device_ordinal < 0 ? Device::cpu() : Device::mi355x(device_ordinal);
        auto h = std::make_unique<Handle>();
        const std::string f(fam);
        if (f == "Llama") { h->family = 0; h->llama = std::make_unique<Model<LlamaWithConfig>>(load_model<LlamaWithConfig>(d, (DType)dtype, dev)); }
        else if (f == "Mistral") { h->family = 1; h->mistral = std::make_unique<Model<MistralWithConfig>>(load_model<MistralWithConfig>(d, (DType)dtype, dev)); }
        else { h->family = 2; h->qwen = std::make_unique<Model<QwenWithConfig>>(load_model<QwenWithConfig>(d, (DType)dtype, dev)); }
        if (family_out) *family_out = h->family;
        *out = h.release();
        return 0;
    });
}

// parse a checkpoint directory on the host only: number of tensors, and for `name` its dtype / shape /
// a byte checksum (sum of bytes mod 2^64) so tests can compare with an independent reader
int flh_checkpoint_probe(const char *dir, const char *name, size_t *n_tensors, int *dtype, int *ndim, int64_t shape[4],
                         uint64_t *byte_sum) {
    return guard([&] {
        Checkpoint ck(dir);
        if (n_tensors) *n_tensors = ck.tensors.size();
        if (name) {
            auto it = ck.tensors.find(name);
            if (it == ck.tensors.end()) throw Error(FL_ERR_MISSING_TENSOR, std::string("cannot find tensor ") + name);
            const Tensor &t = it->second;
            *dtype = (int)t.dtype; *ndim = (int)t.shape.size();
            size_t cnt = 1;
            for (size_t i = 0; i < t.shape.size() && i < 4; i++) { shape[i] = t.shape[i]; cnt *= (size_t)t.shape[i]; }
            const size_t bytes = cnt * (t.dtype == DType::F32 ? 4 : 2);
            uint64_t sum = 0;
            const unsigned char *p = static_cast<const unsigned char *>(t.data);
            for (size_t i = 0; i < bytes; i++) sum += p[i];
            *byte_sum = sum;
        }
        return 0;
    });
}

void flh_model_destroy(void *h) { delete static_cast<Handle *>(h); }

// Model<M>::generate on token ids (mod.rs:363-463)
int flh_generate(void *hv, const uint32_t *prompt, size_t T, size_t max_tokens, float temperature, int64_t eos,
                 uint32_t *out_tokens, size_t *n_out, size_t *forwards) {
    return guard([&] {
        Handle *h = static_cast<Handle *>(hv);
        std::vector<uint32_t> p(prompt, prompt + T), r;
        std::optional<uint32_t> e = eos >= 0 ? std::optional<uint32_t>((uint32_t)eos) : std::nullopt;
        size_t fw = 0;
        if (h->llama) { r = h->llama->generate_ids(p, max_tokens, temperature, e); fw = h->llama->forwards; }
        else if (h->mistral) { r = h->mistral->generate_ids(p, max_tokens, temperature, e); fw = h->mistral->forwards; }
        else { r = h->qwen->generate_ids(p, max_tokens, temperature, e); fw = h->qwen->forwards; }
        for (size_t i = 0; i < r.size(); i++) out_tokens[i] = r[i];
        *n_out = r.size();
        if (forwards) *forwards = fw;
        return 0;
    });
}

// ModelWrapper::generate_stream / generate_tokens_inner on token ids (mod.rs:137-238, 268-340): a stream of its own on the shared
// model -- fresh cache, seed-0 sampler, one callback per token; the callback returning 0 is the dropped receiver.
// Thread-safe for concurrent calls on one handle (each call owns its cache).
int flh_generate_stream(void *hv, const uint32_t *prompt, size_t T, size_t max_tokens, float temperature, int64_t eos,
                        int (*on_token)(uint32_t token, void *user), void *user, size_t *forwards) {
    return guard([&] {
        const Handle *h = static_cast<const Handle *>(hv);
        if (!on_token) throw Error(FL_ERR_BAD_ARGUMENT, "null token callback");
        std::vector<uint32_t> p(prompt, prompt + T);
        std::optional<uint32_t> e = eos >= 0 ? std::optional<uint32_t>((uint32_t)eos) : std::nullopt;
        auto cb = [&](uint32_t t) { return on_token(t, user) != 0; };
        size_t fw = 0;
        if (h->llama) fw = h->llama->generate_stream_ids(p, max_tokens, temperature, e, cb);
        else if (h->mistral) fw = h->mistral->generate_stream_ids(p, max_tokens, temperature, e, cb);
        else fw = h->qwen->generate_stream_ids(p, max_tokens, temperature, e, cb);
        if (forwards) *forwards = fw;
        return 0;
    });
}

// LogitsProcessor::new(seed, temperature, None) / .sample (mod.rs:373-374,425-428); has_temperature 0 = None
int flh_logits_processor_new(uint64_t seed, int has_temperature, double temperature, void **out) {
    return guard([&] {
        *out = new LogitsProcessor(seed, has_temperature ? std::optional<double>(temperature) : std::nullopt);
        return 0;
    });
}
int flh_logits_processor_sample(void *lp, const float *logits, size_t n, uint32_t *token_out, uint64_t *draws_out) {
    return guard([&] {
        LogitsProcessor *p = static_cast<LogitsProcessor *>(lp);
        *token_out = p->sample(logits, n);
        if (draws_out) *draws_out = p->draws();
        return 0;
    });
}
void flh_logits_processor_free(void *lp) { delete static_cast<LogitsProcessor *>(lp); }

// ModelInitializer::forward through the model's own cache object (trait-level call)
int flh_forward(void *hv, const uint32_t *ids, size_t T, size_t pos, float *logits_out, size_t *n_logits) {
    return guard([&] {
        Handle *h = static_cast<Handle *>(hv);
        Tensor in = Tensor::from_ids(std::vector<uint32_t>(ids, ids + T)), lg;
        if (h->llama) lg = h->llama->model.forward(in, pos, h->llama->cache);
        else if (h->mistral) lg = h->mistral->model.forward(in, pos, h->mistral->cache);
        else lg = h->qwen->model.forward(in, pos, h->qwen->cache);
        const size_t n = (size_t)lg.elem_count();
        std::memcpy(logits_out, lg.f32(), n * sizeof(float));
        if (n_logits) *n_logits = n;
        return 0;
    });
}

size_t flh_cache_offset(void *hv) {
    Handle *h = static_cast<Handle *>(hv);
    if (h->llama) return h->llama->cache.get_offset();
    if (h->mistral) return h->mistral->cache.get_offset();
    return h->qwen->cache.get_offset();
}

void flh_cache_reset(void *hv) {           // ModelCache::reset (+ a fresh cache object for Llama)
    Handle *h = static_cast<Handle *>(hv);
    if (h->llama) h->llama->cache = LlamaWithConfig::initialize_cache(h->llama->device, h->llama->dtype);
    else if (h->mistral) h->mistral->cache.reset();
    else h->qwen->cache.reset();
}

// StreamBatcher (fastllm_host.hpp): the reference's concurrent streams (mod.rs:137-238) on one batched decode loop
namespace {
struct BatcherHandle { std::unique_ptr<StreamBatcher> b; };
std::shared_ptr<detail::ModelHandle> model_of(Handle *h) {
    if (h->llama) return h->llama->model.model;
    if (h->mistral) return h->mistral->model.model;
    return h->qwen->model.model;
}
}  // namespace
int flh_batcher_create(void *hv, size_t slots, size_t max_seq, size_t chunk, void **out) {
    return guard([&] {
        if (!hv || !out) throw Error(FL_ERR_BAD_ARGUMENT, "null argument");
        auto bh = std::make_unique<BatcherHandle>();
        Handle *h = static_cast<Handle *>(hv);
        bh->b = std::make_unique<StreamBatcher>(model_of(h), slots, max_seq, chunk, !h->llama && pos_mode() == PosMode::Reference);
        *out = bh.release();
        return 0;
    });
}
// ... on an fl_model the caller already holds (a reference is taken): a host that built the model through the C ABI itself
int flh_batcher_create_on(void *fl_model_ptr, size_t slots, size_t max_seq, size_t chunk, int counter_positions, void **out) {
    return guard([&] {
        if (!fl_model_ptr || !out) throw Error(FL_ERR_BAD_ARGUMENT, "null argument");
        fl_model *m = static_cast<fl_model *>(fl_model_ptr);
        fl_model_retain(m);
        auto mh = std::make_shared<detail::ModelHandle>(m);          // (releases the reference)
        auto bh = std::make_unique<BatcherHandle>();
        bh->b = std::make_unique<StreamBatcher>(mh, slots, max_seq, chunk, counter_positions != 0);
        *out = bh.release();
        return 0;
    });
}
int flh_batcher_submit(void *bv, const uint32_t *prompt, size_t T, size_t max_tokens, float temperature, int64_t eos,
                       int (*on_token)(uint64_t id, uint32_t token, void *user), void (*on_done)(uint64_t id, size_t n_tokens, void *user),
                       void *user, uint64_t *id_out) {
    return guard([&] {
        if (!bv || !on_token || (!prompt && T)) throw Error(FL_ERR_BAD_ARGUMENT, "null argument");
        auto *bh = static_cast<BatcherHandle *>(bv);
        auto id = std::make_shared<uint64_t>(0);
        std::optional<uint32_t> e = eos >= 0 ? std::optional<uint32_t>((uint32_t)eos) : std::nullopt;
        StreamBatcher::OnDone done;
        if (on_done) done = [=](size_t n) { on_done(*id, n, user); };
        *id = bh->b->submit(std::vector<uint32_t>(prompt, prompt + T), max_tokens, temperature, e,
                            [=](uint32_t t) { return on_token(*id, t, user) != 0; }, done);
        if (id_out) *id_out = *id;
        return 0;
    });
}
int flh_batcher_run(void *bv, size_t *batch_steps, size_t *prefills) {
    return guard([&] {
        if (!bv) throw Error(FL_ERR_BAD_ARGUMENT, "null argument");
        auto *bh = static_cast<BatcherHandle *>(bv);
        bh->b->run();
        if (batch_steps) *batch_steps = bh->b->batch_steps;
        if (prefills) *prefills = bh->b->prefills;
        return 0;
    });
}
void flh_batcher_destroy(void *bv) { delete static_cast<BatcherHandle *>(bv); }

}  // extern "C"
