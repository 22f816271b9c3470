// fastllm_host.hpp -- host-side mirror of FastLLM's model-trait surface for the causal-LM forward
// path, written in C++ over the C ABI (include/fastllm_mi355x.h).
//
// The reference's host language is Rust and there is no Rust toolchain in the build image, so the
// layer a Rust maintainer would write as `impl ModelInitializer for Mi355xWithConfig`
// (INTEGRATION.md shows that binding) is mirrored here 1:1 in C++: same type names, same argument
// meaning, same error behaviour, so the parity tests read like the reference's own unit tests.
//
//   reference item                                   mirror
//   ------------------------------------------------ -----------------------------------------
//   trait ModelInitializer  (model_initializer.rs:6-22)   static initialize_model / initialize_cache, forward
//   trait ModelArchitecture (model_initializer.rs:24-27)  static get_family / supports_architecture
//   trait ModelCache, CommonCache (cache.rs:5-42)          struct ModelCache, CommonCache
//   BaseModelConfig + ModelConfigValidation (config.rs)    struct BaseModelConfig
//   llama::ConfigFile, LlamaCache, LlamaWithConfig         same names (llama.rs:18-159)
//   mistral::ConfigFile, MistralCache, MistralWithConfig   same names (mistral.rs:16-247)
//   QwenCache, QwenWithConfig                              same names (qwen.rs:13-184)
//   generate_stream / generate_tokens_inner (mod.rs:137-340)   Model<M>::generate_stream_ids (per-stream cache, token callback)
//   Model<M>::generate (mod.rs:363-463)                    Model<M>::generate_ids (token ids in/out;
//                                                          the tokenizer is outside the hot path)
//
// Errors: anyhow::Result -> fastllm::Error (exception, carries the fl_status code); config
// violations that `assert!`/`expect` in the reference (mistral.rs:109-127, qwen.rs:32-37) throw
// fastllm::Panic.  There is no CPU fallback: Device::Cpu is rejected by initialize_model.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/fastllm_mi355x.h"

namespace fastllm {

// ------------------------------------------------------------------------------------ errors
struct Error : std::runtime_error {          // anyhow::Error
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
struct Panic : std::logic_error {            // assert!/expect in the reference
    explicit Panic(const std::string &m) : std::logic_error(m) {}
};
inline void check(int rc, const char *ctx) {
    if (rc != FL_OK) throw Error(rc, std::string(ctx) + ": " + fl_last_error());
}

// ------------------------------------------------------------------------------------ candle stand-ins
enum class DType { F32 = FL_DTYPE_F32, BF16 = FL_DTYPE_BF16, F16 = FL_DTYPE_F16 };   // candle_core::DType

struct Device {                               // candle_core::Device
    enum Kind { Cpu, Mi355x } kind = Cpu;
    std::vector<int32_t> ordinals;            // one GPU, or the TP group driven by this process
    static Device cpu() { return Device{}; }
    static Device mi355x(int ordinal = 0) { Device d; d.kind = Mi355x; d.ordinals = {ordinal}; return d; }
    static Device mi355x_tp(std::vector<int32_t> ords) { Device d; d.kind = Mi355x; d.ordinals = std::move(ords); return d; }
    // main.rs:81-97 on linux: Device::cuda_if_available(0) -> the accelerator when present, else Cpu
    static Device cuda_if_available(int ordinal) {
        int n = 0; fl_device_count(&n);
        return n > ordinal ? mi355x(ordinal) : cpu();
    }
    bool is_cpu() const { return kind == Cpu; }
};

struct Tensor {                               // a borrowed view, like the HashMap<String, Tensor> entries
    DType dtype = DType::F32;
    std::vector<int64_t> shape;
    const void *data = nullptr;
    int32_t device = -1;                      // -1 host
    std::shared_ptr<void> owner;              // keeps host storage alive when the tensor owns it

    static Tensor from_ids(const std::vector<uint32_t> &ids) {          // Tensor::new(ids, dev).reshape((1,T))
        auto buf = std::make_shared<std::vector<uint32_t>>(ids);
        Tensor t; t.dtype = DType::F32; t.shape = {1, (int64_t)ids.size()}; t.data = buf->data(); t.owner = buf;
        return t;
    }
    std::pair<int64_t, int64_t> dims2() const {
        if (shape.size() != 2) throw Error(FL_ERR_SHAPE_MISMATCH, "unexpected rank, expected 2");
        return {shape[0], shape[1]};
    }
    // logits.get(0)?.flatten_all()?.to_vec1::<f32>()  (mod.rs:421; LogitsProcessor::sample)
    const float *f32() const { return static_cast<const float *>(data); }
    int64_t elem_count() const { int64_t n = 1; for (auto s : shape) n *= s; return n; }
};
using TensorMap = std::unordered_map<std::string, Tensor>;

// ------------------------------------------------------------------------------------ cache.rs
struct ModelCache {                           // cache.rs:5-10
    virtual void increment_offset() = 0;
    virtual void reset() = 0;
    virtual size_t get_offset() const = 0;
    virtual ~ModelCache() = default;
};
struct CommonCache : ModelCache {             // cache.rs:13-42
    size_t seqlen_offset = 0;
    void increment_offset() override { seqlen_offset += 1; }
    void reset() override { seqlen_offset = 0; }
    size_t get_offset() const override { return seqlen_offset; }
};

// ------------------------------------------------------------------------------------ minimal JSON (config.json)
namespace detail {
struct JsonFlat {                             // top-level object: numbers, strings, null/bool; nested values skipped
    std::unordered_map<std::string, std::string> raw;
    static void skip_ws(const std::string &s, size_t &i) { while (i < s.size() && std::isspace((unsigned char)s[i])) i++; }
    static std::string parse_string(const std::string &s, size_t &i) {
        std::string out; i++;
        while (i < s.size() && s[i] != '"') { if (s[i] == '\\' && i + 1 < s.size()) i++; out += s[i++]; }
        if (i >= s.size()) throw Error(FL_ERR_BAD_CONFIG, "unterminated string in config JSON");
        i++; return out;
    }
    static void skip_value(const std::string &s, size_t &i) {
        int depth = 0;
        while (i < s.size()) {
            char c = s[i];
            if (c == '"') { parse_string(s, i); continue; }
            if (c == '{' || c == '[') depth++;
            else if (c == '}' || c == ']') { if (depth == 0) return; depth--; }
            else if (c == ',' && depth == 0) return;
            i++;
        }
    }
    explicit JsonFlat(const std::string &s) {
        size_t i = 0; skip_ws(s, i);
        if (i >= s.size() || s[i] != '{') throw Error(FL_ERR_BAD_CONFIG, "config JSON must be an object");
        i++;
        while (true) {
            skip_ws(s, i);
            if (i < s.size() && s[i] == '}') break;
            if (i >= s.size() || s[i] != '"') throw Error(FL_ERR_BAD_CONFIG, "malformed config JSON");
            std::string key = parse_string(s, i);
            skip_ws(s, i);
            if (i >= s.size() || s[i] != ':') throw Error(FL_ERR_BAD_CONFIG, "malformed config JSON (missing ':')");
            i++; skip_ws(s, i);
            size_t b = i;
            if (s[i] == '"') { std::string v = parse_string(s, i); raw[key] = "\"" + v; }
            else { skip_value(s, i); std::string v = s.substr(b, i - b); while (!v.empty() && std::isspace((unsigned char)v.back())) v.pop_back(); raw[key] = v; }
            skip_ws(s, i);
            if (i < s.size() && s[i] == ',') { i++; continue; }
            if (i < s.size() && s[i] == '}') break;
            throw Error(FL_ERR_BAD_CONFIG, "malformed config JSON");
        }
    }
    bool has(const std::string &k) const { auto it = raw.find(k); return it != raw.end() && it->second != "null"; }
    double num(const std::string &k) const {                  // serde: missing non-Option field is an error
        auto it = raw.find(k);
        if (it == raw.end() || it->second == "null") throw Error(FL_ERR_BAD_CONFIG, "missing field `" + k + "`");
        char *end = nullptr; double v = std::strtod(it->second.c_str(), &end);
        if (end == it->second.c_str()) throw Error(FL_ERR_BAD_CONFIG, "invalid type for field `" + k + "`");
        return v;
    }
    std::optional<double> opt(const std::string &k) const { return has(k) ? std::optional<double>(num(k)) : std::nullopt; }
    // serde into usize: a non-negative integer; negative, fractional or out-of-range numbers are a type error, never a cast
    size_t usize(const std::string &k) const {
        const double v = num(k);
        if (!(v >= 0.0) || !(v <= 9007199254740992.0) || (double)(uint64_t)v != v) throw Error(FL_ERR_BAD_CONFIG, "invalid value for field `" + k + "`: expected usize");
        return (size_t)v;
    }
    std::optional<size_t> opt_usize(const std::string &k) const { return has(k) ? std::optional<size_t>(usize(k)) : std::nullopt; }
    std::optional<std::string> str(const std::string &k) const {
        auto it = raw.find(k);
        if (it == raw.end() || it->second.empty() || it->second[0] != '"') return std::nullopt;
        return it->second.substr(1);
    }
};
}  // namespace detail

// ------------------------------------------------------------------------------------ config.rs
struct BaseModelConfig {                      // config.rs:6-18 (also llama::ConfigFile llama.rs:18-29, mistral::ConfigFile mistral.rs:80-92)
    size_t hidden_size = 0, intermediate_size = 0, vocab_size = 0, num_hidden_layers = 0, num_attention_heads = 0;
    std::optional<size_t> num_key_value_heads;
    double rms_norm_eps = 0;
    std::optional<double> rope_theta;
    std::optional<size_t> max_position_embeddings;
    std::optional<size_t> sliding_window;
    std::optional<std::string> torch_dtype;

    static BaseModelConfig from_json(const std::string &text) {          // serde_json::from_str (huggingface.rs:78-79)
        detail::JsonFlat j(text);
        BaseModelConfig c;
        c.hidden_size = j.usize("hidden_size");
        c.intermediate_size = j.usize("intermediate_size");
        c.vocab_size = j.usize("vocab_size");
        c.num_hidden_layers = j.usize("num_hidden_layers");
        c.num_attention_heads = j.usize("num_attention_heads");
        c.num_key_value_heads = j.opt_usize("num_key_value_heads");
        c.rms_norm_eps = j.num("rms_norm_eps");
        c.rope_theta = j.opt("rope_theta");
        c.max_position_embeddings = j.opt_usize("max_position_embeddings");
        c.sliding_window = j.opt_usize("sliding_window");
        c.torch_dtype = j.str("torch_dtype");
        return c;
    }
    // ModelConfigValidation (config.rs:31-54)
    size_t validate_head_dimensions() const {
        if (num_attention_heads == 0) throw Error(FL_ERR_BAD_CONFIG, "hidden_size must be divisible by num_attention_heads");
        size_t head_dim = hidden_size / num_attention_heads;
        if (head_dim * num_attention_heads != hidden_size) throw Error(FL_ERR_BAD_CONFIG, "hidden_size must be divisible by num_attention_heads");
        if (head_dim % 2 != 0) throw Error(FL_ERR_BAD_CONFIG, "head_dim must be even for RoPE embeddings");
        return head_dim;
    }
    void validate_gqa_config() const {
        if (num_key_value_heads && (*num_key_value_heads == 0 || num_attention_heads % *num_key_value_heads != 0))
            throw Error(FL_ERR_BAD_CONFIG, "num_attention_heads must be divisible by num_key_value_heads");
    }
    fl_config to_fl(fl_family family, bool qkv_bias) const {
        fl_config c{};
        c.family = family; c.qkv_bias = qkv_bias;
        c.hidden_size = (int64_t)hidden_size; c.intermediate_size = (int64_t)intermediate_size;
        c.vocab_size = (int64_t)vocab_size; c.num_hidden_layers = (int64_t)num_hidden_layers;
        c.num_attention_heads = (int64_t)num_attention_heads;
        c.num_key_value_heads = (int64_t)num_key_value_heads.value_or(0);
        c.max_position_embeddings = (int64_t)max_position_embeddings.value_or(0);
        c.sliding_window = (int64_t)sliding_window.value_or(0);
        c.rms_norm_eps = rms_norm_eps; c.rope_theta = rope_theta.value_or(0.0);
        return c;
    }
};

// ------------------------------------------------------------------------------------ shared plumbing
namespace detail {
struct ModelHandle {                          // owns one fl_model reference; Clone = refcount bump (mod.rs:155)
    fl_model *m = nullptr;
    explicit ModelHandle(fl_model *mm) : m(mm) {}
    ~ModelHandle() { if (m) fl_model_release(m); }
    ModelHandle(const ModelHandle &) = delete;
};
struct CacheHandle {
    fl_cache *c = nullptr;
    ~CacheHandle() { if (c) fl_cache_destroy(c); }
};

inline size_t default_max_seq(const fl_model *m) {
    fl_model_info info; check(fl_model_get_info(m, &info), "fl_model_get_info");
    size_t cap = 4096;                                         // KV capacity of one request
    if (const char *e = std::getenv("FASTLLM_MAX_SEQ")) cap = (size_t)std::strtoull(e, nullptr, 10);
    return std::min<size_t>(cap, (size_t)info.cfg.max_position_embeddings);
}

inline std::shared_ptr<ModelHandle> create(const fl_config &cfg, const TensorMap &tensors, DType dtype, const Device &device) {
    if (device.is_cpu())
        throw Error(FL_ERR_NO_DEVICE, "the MI355X backend has no CPU path: pass Device::mi355x(..) (main.rs:81-97 picks the accelerator)");
    std::vector<fl_tensor> arr;
    arr.reserve(tensors.size());
    for (auto &kv : tensors) {
        fl_tensor t{};
        t.name = kv.first.c_str(); t.dtype = (int32_t)kv.second.dtype; t.ndim = (int32_t)kv.second.shape.size();
        if (t.ndim > 4) throw Error(FL_ERR_SHAPE_MISMATCH, "tensor " + kv.first + " has rank > 4");
        for (int i = 0; i < t.ndim; i++) t.shape[i] = kv.second.shape[i];
        t.data = kv.second.data; t.device = kv.second.device;
        arr.push_back(t);
    }
    fl_parallel par{};
    par.mode = device.ordinals.size() > 1 ? FL_TP_SINGLE_PROCESS : FL_TP_NONE;
    par.tp_size = (int32_t)std::max<size_t>(1, device.ordinals.size());
    par.device_ids = device.ordinals.data(); par.n_device_ids = (int32_t)device.ordinals.size();
    fl_model *m = nullptr;
    check(fl_model_create(&cfg, arr.data(), arr.size(), (int32_t)dtype, &par, &m), "Failed to initialize model");
    return std::make_shared<ModelHandle>(m);
}

inline Tensor run_forward(fl_model *m, fl_cache *c, const Tensor &input, size_t pos) {
    auto [b, t] = input.dims2();
    if (b != 1) throw Error(FL_ERR_BAD_ARGUMENT, "batch size must be 1 (mod.rs:283-291)");
    fl_model_info info; check(fl_model_get_info(m, &info), "fl_model_get_info");
    auto buf = std::make_shared<std::vector<float>>((size_t)info.cfg.vocab_size);
    check(fl_forward(m, c, static_cast<const uint32_t *>(input.data), (size_t)t, pos, buf->data()), "Model forward pass failed");
    Tensor out; out.dtype = DType::F32; out.data = buf->data(); out.owner = buf;
    return out;
}
}  // namespace detail

enum class PosMode { Reference, Tokens };
// Reference (default): bug-compatible with the reference -- Mistral/Qwen pass a per-call counter as the
// RoPE offset (mistral.rs:226,234; qwen.rs:142-143).  Tokens: the offset is the token position.
inline PosMode pos_mode() {
    const char *e = std::getenv("FASTLLM_POS_MODE");
    return (e && std::string(e) == "tokens") ? PosMode::Tokens : PosMode::Reference;
}

// ------------------------------------------------------------------------------------ llama.rs
struct LlamaCache : ModelCache {              // llama.rs:62-92: wraps candle's Cache{cos,sin,kvs} + an unused counter
    std::shared_ptr<detail::CacheHandle> inner;
    size_t seqlen_offset = 0;
    void increment_offset() override { seqlen_offset += 1; }
    void reset() override { seqlen_offset = 0; }
    size_t get_offset() const override { return seqlen_offset; }
};

struct LlamaWithConfig {
    using Config = BaseModelConfig;           // llama::ConfigFile (llama.rs:18-29): same fields minus sliding_window
    using Cache = LlamaCache;
    std::shared_ptr<detail::ModelHandle> model;

    static std::shared_ptr<detail::ModelHandle> &last() { static std::shared_ptr<detail::ModelHandle> p; return p; }

    // llama.rs:98-123.  From<ConfigFile> (llama.rs:31-50): kv heads default to heads, theta 1e4, max pos 4096.
    static std::pair<LlamaWithConfig, LlamaCache> initialize_model(const Config &config, const TensorMap &tensors, DType dtype, const Device &device) {
        LlamaWithConfig self;
        self.model = detail::create(config.to_fl(FL_FAMILY_LLAMA, false), tensors, dtype, device);
        last() = self.model;
        return {self, new_cache(self.model)};
    }
    // llama.rs:125-145 builds the cache from hard-coded TinyLlama dims because the associated fn cannot see
    // the model (quirk C.2).  Here it is derived from the most recently initialised model of this process
    // (the reference serves one model per process, main.rs:128).
    static LlamaCache initialize_cache(const Device &, DType) {
        if (!last()) throw Error(FL_ERR_BAD_ARGUMENT, "Failed to initialize model cache: no model initialised");
        return new_cache(last());
    }
    // llama.rs:147-149: self.model.forward(input, pos, &mut cache.inner) -- the caller's token position is used
    Tensor forward(const Tensor &input, size_t pos, Cache &cache) const {
        Tensor logits = detail::run_forward(model->m, cache.inner->c, input, pos);
        logits.shape = {1, logits.owner ? (int64_t)static_cast<std::vector<float> *>(logits.owner.get())->size() : 0};   // [1, V] f32
        return logits;
    }
    static const char *get_family() { return "Llama"; }                                             // llama.rs:153
    static bool supports_architecture(const std::string &a) { return a == "LlamaForCausalLM"; }     // llama.rs:157-159

   private:
    static LlamaCache new_cache(const std::shared_ptr<detail::ModelHandle> &m) {
        LlamaCache c; c.inner = std::make_shared<detail::CacheHandle>();
        check(fl_cache_create(m->m, detail::default_max_seq(m->m), &c.inner->c), "Failed to initialize model cache");
        return c;
    }
};

// ------------------------------------------------------------------------------------ mistral.rs / qwen.rs
// The reference keeps KV inside the candle model (RefCell) and the adapter cache is only a counter
// (mistral.rs:16-47, qwen.rs:59-87).  Here KV is a caller-owned fl_cache attached to the counter object,
// so concurrent streams on one model are sound (quirk C.7).
struct CounterCache : ModelCache {
    size_t seqlen_offset = 0;
    std::shared_ptr<detail::CacheHandle> kv;  // created lazily by forward (initialize_cache cannot see the model)
    void increment_offset() override { seqlen_offset += 1; }
    void reset() override { seqlen_offset = 0; }
    size_t get_offset() const override { return seqlen_offset; }
};
using MistralCache = CounterCache;
using QwenCache = CounterCache;

template <fl_family FAMILY, bool BIAS>
struct CounterFamily {
    using Config = BaseModelConfig;
    using Cache = CounterCache;
    std::shared_ptr<detail::ModelHandle> model;

    static size_t get_head_dim(size_t hidden_size, size_t num_attention_heads) {                    // mistral.rs:67-76
        if (num_attention_heads == 0 || (hidden_size / num_attention_heads) * num_attention_heads != hidden_size)
            throw Panic("hidden_size must be divisible by num_attention_heads");
        return hidden_size / num_attention_heads;
    }
    static void validate(const Config &cf) {                                                        // mistral.rs:93-127, qwen.rs:30-37
        size_t head_dim = get_head_dim(cf.hidden_size, cf.num_attention_heads);
        size_t kv = cf.num_key_value_heads.value_or(cf.num_attention_heads);
        if (kv == 0 || cf.num_attention_heads % kv != 0) throw Panic("num_attention_heads must be divisible by num_key_value_heads");
        if (head_dim % 2 != 0) throw Panic("head_dim must be even for RoPE embeddings");
    }
    static std::pair<CounterFamily, Cache> initialize_model(const Config &config, const TensorMap &tensors, DType dtype, const Device &device) {
        validate(config);
        CounterFamily self;
        self.model = detail::create(config.to_fl(FAMILY, BIAS), tensors, dtype, device);
        return {self, Cache{}};
    }
    static Cache initialize_cache(const Device &, DType) { return Cache{}; }                        // mistral.rs:202-204, qwen.rs:119-121
    // mistral.rs:206-236 / qwen.rs:123-151: `_pos` is ignored; offset 0 clears the KV cache; the model is run
    // at the per-call counter; the counter then advances by ONE per call (quirk C.1).
    Tensor forward(const Tensor &input, size_t pos, Cache &cache) const {
        if (!cache.kv) {
            cache.kv = std::make_shared<detail::CacheHandle>();
            check(fl_cache_create(model->m, detail::default_max_seq(model->m), &cache.kv->c), "Failed to initialize model cache");
        }
        if (cache.seqlen_offset == 0) fl_cache_reset(cache.kv->c);                                  // clear_kv_cache()
        const size_t rope_offset = pos_mode() == PosMode::Reference ? cache.seqlen_offset : pos;
        Tensor logits = detail::run_forward(model->m, cache.kv->c, input, rope_offset);
        logits.shape = {1, 1, (int64_t)static_cast<std::vector<float> *>(logits.owner.get())->size()};   // [1,1,V]
        cache.increment_offset();
        return logits;
    }
};

struct MistralWithConfig : CounterFamily<FL_FAMILY_MISTRAL, false> {
    MistralWithConfig() = default;
    MistralWithConfig(const CounterFamily &b) : CounterFamily(b) {}
    static std::pair<MistralWithConfig, MistralCache> initialize_model(const Config &c, const TensorMap &t, DType d, const Device &dev) {
        auto r = CounterFamily::initialize_model(c, t, d, dev); return {MistralWithConfig(r.first), r.second};
    }
    static const char *get_family() { return "Mistral"; }                                           // mistral.rs:240
    static bool supports_architecture(const std::string &a) { return a == "MistralForCausalLM"; }   // mistral.rs:244-246
};

struct QwenWithConfig : CounterFamily<FL_FAMILY_QWEN2, true> {
    QwenWithConfig() = default;
    QwenWithConfig(const CounterFamily &b) : CounterFamily(b) {}
    // qwen.rs:30-37: validate_head_dimensions().expect(..) / validate_gqa_config().expect(..) -> panic
    static std::pair<QwenWithConfig, QwenCache> initialize_model(const Config &c, const TensorMap &t, DType d, const Device &dev) {
        try { c.validate_head_dimensions(); } catch (const Error &e) { throw Panic(std::string("Invalid head dimensions: ") + e.what()); }
        try { c.validate_gqa_config(); } catch (const Error &e) { throw Panic(std::string("Invalid GQA configuration: ") + e.what()); }
        auto r = CounterFamily::initialize_model(c, t, d, dev); return {QwenWithConfig(r.first), r.second};
    }
    static const char *get_family() { return "Qwen"; }                                              // qwen.rs:174
    static bool supports_architecture(const std::string &a) {                                       // qwen.rs:178-183
        return a == "Qwen2ForCausalLM" || a == "Qwen2_5_VLForConditionalGeneration";
    }
};

// ------------------------------------------------------------------------------------ mod.rs: Model<M>
inline uint32_t argmax_last(const float *v, size_t n) {      // LogitsProcessor ArgMax: max_by(total_cmp) -> last max
    size_t best = 0;
    for (size_t i = 1; i < n; i++) if (!(v[i] < v[best])) best = i;
    return (uint32_t)best;
}

// rand 0.8 `StdRng` as LogitsProcessor seeds it: ChaCha12, key = SeedableRng::seed_from_u64 (PCG32 XSH-RR
// expansion of the u64 into eight little-endian words), 64-bit block counter from 0, stream id 0, consumed
// one u32 word at a time.
class StdRng {
   public:
    explicit StdRng(uint64_t seed) {
        uint64_t st = seed;
        for (auto &k : key_) {
            st = st * 6364136223846793005ull + 11634580027462260723ull;
            const uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27), rot = (uint32_t)(st >> 59);
            k = (xs >> rot) | (xs << ((32 - rot) & 31));
        }
    }
    uint32_t next_u32() {
        if ((word_ & 15) == 0) block(word_ >> 4);
        return buf_[word_++ & 15];
    }
    uint64_t words_consumed() const { return word_; }

   private:
    static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    static void qr(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d) {
        a += b; d ^= a; d = rotl(d, 16); c += d; b ^= c; b = rotl(b, 12);
        a += b; d ^= a; d = rotl(d, 8);  c += d; b ^= c; b = rotl(b, 7);
    }
    void block(uint64_t counter) {
        uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 8; i++) s[4 + i] = key_[i];
        s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = s[15] = 0;
        uint32_t x[16];
        std::memcpy(x, s, sizeof x);
        for (int r = 0; r < 12; r += 2) {
            qr(x[0], x[4], x[8], x[12]); qr(x[1], x[5], x[9], x[13]); qr(x[2], x[6], x[10], x[14]); qr(x[3], x[7], x[11], x[15]);
            qr(x[0], x[5], x[10], x[15]); qr(x[1], x[6], x[11], x[12]); qr(x[2], x[7], x[8], x[13]); qr(x[3], x[4], x[9], x[14]);
        }
        for (int i = 0; i < 16; i++) buf_[i] = x[i] + s[i];
    }
    uint32_t key_[8], buf_[16];
    uint64_t word_ = 0;
};

// candle_transformers::generation::LogitsProcessor as the reference builds it:
// LogitsProcessor::new(Default::default(), Some(temperature as f64), None) (mod.rs:157-158,373-374).
// This is the reference's own host-side shape (logits on the host, mod.rs:421-428); the device-side
// equivalent is fl_forward_sample / fl_decode_sample.
class LogitsProcessor {
   public:
    LogitsProcessor(uint64_t seed, std::optional<double> temperature) : rng_(seed) {
        if (temperature && !(*temperature < 1e-7)) { sampling_ = true; temperature_ = *temperature; }
    }
    bool is_argmax() const { return !sampling_; }
    double temperature() const { return temperature_; }
    uint64_t draws() const { return rng_.words_consumed(); }
    uint32_t sample(const float *logits, size_t n) {
        if (n == 0) throw Error(FL_ERR_BAD_ARGUMENT, "empty logits");
        if (!sampling_) return argmax_last(logits, n);
        // prs = softmax_last_dim(logits / temperature): `Tensor / f64` is affine(1/t, 0) in the tensor's dtype
        const float mul = (float)(1.0 / temperature_);
        prs_.resize(n);
        float mx = -INFINITY;
        for (size_t i = 0; i < n; i++) { prs_[i] = logits[i] * mul; if (prs_[i] > mx) mx = prs_[i]; }
        float sum = 0.f;
        for (size_t i = 0; i < n; i++) { prs_[i] = std::exp(prs_[i] - mx); sum += prs_[i]; }
        for (size_t i = 0; i < n; i++) prs_[i] /= sum;
        // WeightedIndex::new: running f32 total, left to right; weights must be >= 0 and not all zero
        float total = 0.f;
        for (size_t i = 0; i < n; i++) {
            if (!(prs_[i] >= 0.f)) throw Error(FL_ERR_BAD_ARGUMENT, "Failed to sample next token: invalid weight");
            total = i == 0 ? prs_[0] : total + prs_[i];
        }
        if (total == 0.f) throw Error(FL_ERR_BAD_ARGUMENT, "Failed to sample next token: all weights zero");
        // UniformFloat::<f32>::new(0, total) and one sample
        float scale = total;
        const float max_rand = 1.0f - 1.1920929e-07f;
        while (mul_rn(scale, max_rand) >= total) { uint32_t b; std::memcpy(&b, &scale, 4); b -= 1; std::memcpy(&scale, &b, 4); }
        const uint32_t bits = (rng_.next_u32() >> 9) | 0x3f800000u;
        float v12; std::memcpy(&v12, &bits, 4);
        const float chosen = mul_rn(v12 - 1.0f, scale);
        // partition_point(|w| w <= chosen) over the n-1 cumulative weights
        float cum = prs_[0];
        size_t idx = 0;
        while (idx < n - 1 && cum <= chosen) { idx++; cum += prs_[idx]; }
        return (uint32_t)idx;
    }

   private:
    static float mul_rn(float a, float b) { volatile float r = a * b; return r; }     // a product, never fused with an add
    StdRng rng_;
    bool sampling_ = false;
    double temperature_ = 0;
    std::vector<float> prs_;
};

template <class M>
struct Model {                                // mod.rs:342-361
    M model;
    Device device;
    typename M::Cache cache;
    DType dtype = DType::BF16;                // mod.rs:359
    size_t forwards = 0;                      // forwards executed by the last generate (incl. the wasted one, C.5)

    Model(M m, Device d, typename M::Cache c) : model(std::move(m)), device(std::move(d)), cache(std::move(c)) {}

    // Model::generate (mod.rs:363-463) on token ids.  eos = tokenizer.token_to_id("</s>") (mod.rs:431).
    std::vector<uint32_t> generate_ids(const std::vector<uint32_t> &prompt, size_t max_tokens, float temperature,
                                       std::optional<uint32_t> eos = std::nullopt) {
        cache = M::initialize_cache(device, dtype);                       // mod.rs:370
        LogitsProcessor logits_processor(0, (double)temperature);         // mod.rs:373-374 (Default::default() seed)
        if (prompt.empty()) throw Error(FL_ERR_BAD_ARGUMENT, "Tokenization error: empty prompt");
        Tensor input = Tensor::from_ids(prompt);                           // mod.rs:386-394
        std::vector<uint32_t> output_ids;
        size_t pos = 0;
        forwards = 0;
        Tensor logits = model.forward(input, pos, cache); forwards++;      // mod.rs:402-405
        pos += prompt.size();                                              // mod.rs:408
        for (size_t i = 0; i < max_tokens; i++) {                          // mod.rs:411
            const uint32_t next = logits_processor.sample(logits.f32(), (size_t)logits.elem_count());   // mod.rs:421-428
            if (eos && next == *eos) break;                                // mod.rs:431-436
            output_ids.push_back(next);                                    // mod.rs:438
            Tensor next_input = Tensor::from_ids({next});                  // mod.rs:441-444
            logits = model.forward(next_input, pos, cache); forwards++;    // mod.rs:446-451 (also after the last token)
            pos += 1;                                                      // mod.rs:452
        }
        return output_ids;
    }

    // ModelWrapper::generate_stream + generate_tokens_inner (mod.rs:137-238, 268-340) on token ids: the model is CLONED (a
    // reference-count bump, mod.rs:155), the stream gets a FRESH cache of its own (mod.rs:156: this object's `cache` is not
    // touched, so several streams may run on one model at once, each from its own thread -- the reference spawns a task per
    // stream), a seed-0 LogitsProcessor (mod.rs:157-158), and every sampled token goes to `on_token` before the next forward
    // (mod.rs:323-325); `on_token` returning false is the dropped receiver (`tx.send(..).is_err()` -> break).  EOS stops the
    // stream before the token is emitted (mod.rs:312-316).  Returns the number of forwards executed (the one behind the last
    // emitted token included, as in the reference's loop).
    template <class OnToken>
    size_t generate_stream_ids(const std::vector<uint32_t> &prompt, size_t max_tokens, float temperature,
                               std::optional<uint32_t> eos, OnToken &&on_token) const {
        M shared = model;                                                  // Arc::new(RwLock::new(model.model.clone()))
        typename M::Cache own = M::initialize_cache(device, dtype);       // one cache per stream
        LogitsProcessor logits_processor(0, (double)temperature);
        if (prompt.empty()) throw Error(FL_ERR_BAD_ARGUMENT, "Tokenization error: empty prompt");
        Tensor input = Tensor::from_ids(prompt);                           // mod.rs:283-291
        size_t pos = 0, n_forwards = 0;
        Tensor logits = shared.forward(input, pos, own); n_forwards++;     // mod.rs:296-298
        pos += prompt.size();
        for (size_t i = 0; i < max_tokens; i++) {                          // mod.rs:303
            const uint32_t next = logits_processor.sample(logits.f32(), (size_t)logits.elem_count());   // mod.rs:305-310
            if (eos && next == *eos) break;                                // mod.rs:312-316
            if (!on_token(next)) break;                                    // mod.rs:323-325
            Tensor next_input = Tensor::from_ids({next});                  // mod.rs:328-331
            logits = shared.forward(next_input, pos, own); n_forwards++;   // mod.rs:333-335
            pos += 1;
        }
        return n_forwards;
    }
};

// ------------------------------------------------------------------------------------ concurrent streams, one batched loop
// The reference serves concurrent requests as independent tasks (ModelWrapper::generate_stream spawns one loop per stream,
// mod.rs:137-238), each paying for the whole weight read per token.  StreamBatcher gives the same streams ONE decode loop
// (continuous batching): `slots` caches form an fl_batch; a request waits in a queue until a slot is free, is
// prefilled there alone (fl_forward_sample: its first token), and from then on advances with the others, `chunk` steps per call of
// fl_batch_decode_each -- every stream with its own temperature (a seed-0 sampler of its own, mod.rs:157-158), its own EOS id and
// its own max_tokens.  Per stream the tokens are those of generate_stream_ids with the sampling done on the device (fl_decode_sample):
// sampled, EOS checked before the token is emitted (mod.rs:312-316), the callback's `false` is the dropped receiver (mod.rs:323-325).
// Steps a stream runs past its own end inside a chunk are computed and discarded (its slot's cache is reset on the next admission).
// Not thread-safe: one thread drives submit() / step() (a server would feed it from a channel).
class StreamBatcher {
  public:
    using OnToken = std::function<bool(uint32_t)>;
    using OnDone = std::function<void(size_t)>;                  // tokens handed to on_token

    // counter_positions: the RoPE offset of a stream's k-th forward is k, not its token position -- what Mistral / Qwen streams
    // see in the reference (quirk C.1: mistral.rs:226,234, qwen.rs:142-143; PosMode::Reference); Llama streams and
    // FASTLLM_POS_MODE=tokens pass token positions
    StreamBatcher(std::shared_ptr<detail::ModelHandle> model, size_t slots, size_t max_seq = 0, size_t chunk = 8, bool counter_positions = false)
        : model_(std::move(model)), chunk_(std::max<size_t>(1, chunk)), counter_(counter_positions) {
        if (!model_ || !model_->m) throw Error(FL_ERR_BAD_ARGUMENT, "StreamBatcher: no model");
        if (slots < 1 || slots > 64) throw Error(FL_ERR_BAD_ARGUMENT, "StreamBatcher: 1 ... 64 slots");
        cap_ = max_seq ? max_seq : detail::default_max_seq(model_->m);
        if (cap_ < chunk_ + 1) throw Error(FL_ERR_BAD_ARGUMENT, "StreamBatcher: the caches are shorter than one chunk");
        slots_.resize(slots);
        std::vector<fl_cache *> cs;
        try {
            for (auto &sl : slots_) { check(fl_cache_create(model_->m, cap_, &sl.cache), "fl_cache_create"); cs.push_back(sl.cache); }
            check(fl_batch_create(model_->m, cs.data(), cs.size(), &batch_), "fl_batch_create");
        } catch (...) { release(); throw; }
    }
    ~StreamBatcher() { release(); }
    StreamBatcher(const StreamBatcher &) = delete;

    // a request as generate_stream sees it (token ids in, one callback per token out); returns its id
    uint64_t submit(std::vector<uint32_t> prompt, size_t max_tokens, float temperature, std::optional<uint32_t> eos, OnToken on_token,
                    OnDone on_done = {}) {
        if (prompt.empty()) throw Error(FL_ERR_BAD_ARGUMENT, "Tokenization error: empty prompt");
        if (prompt.size() + 1 > cap_) throw Error(FL_ERR_BAD_ARGUMENT, "the prompt does not fit a slot's cache");
        if (!on_token) throw Error(FL_ERR_BAD_ARGUMENT, "null token callback");
        Request r;
        r.id = ++next_id_; r.prompt = std::move(prompt); r.max_tokens = max_tokens; r.temperature = temperature; r.eos = eos;
        r.on_token = std::move(on_token); r.on_done = std::move(on_done);
        queue_.push_back(std::move(r));
        return next_id_;
    }

    size_t active() const { size_t n = 0; for (auto &sl : slots_) n += sl.active ? 1 : 0; return n; }
    size_t waiting() const { return queue_.size(); }
    size_t batch_steps = 0;                                        // decode steps the batch has run (all slots advance together)
    size_t prefills = 0;

    // admit waiting requests into free slots, then one chunk for everybody; false: nothing active and nothing waiting
    bool step() {
        for (auto &sl : slots_) {
            while (!sl.active && !queue_.empty()) { admit(sl, std::move(queue_.front())); queue_.pop_front(); }
        }
        if (active() == 0) return !queue_.empty();
        const size_t B = slots_.size();
        size_t n = chunk_, want = 0;
        for (auto &sl : slots_)
            if (sl.active) { n = std::min(n, cap_ - sl.pos); want = std::max(want, sl.req.max_tokens - sl.emitted); }
        n = std::min(n, want);                                     // nobody needs more than `want` further tokens
        std::vector<uint32_t> first(B, 0), out(B * n, 0);
        std::vector<size_t> pos(B, 0), n_out(B, 0);
        std::vector<int64_t> eos(B, -1);
        std::vector<fl_sampling> sp(B, fl_sampling{0.0, 0, 0});
        for (size_t i = 0; i < B; i++) {
            Slot &sl = slots_[i];
            if (!sl.active) { fl_cache_reset(sl.cache); continue; }     // an empty slot computes a throw-away row at position 0
            first[i] = sl.tok; pos[i] = counter_ ? sl.calls : sl.pos;
            if (sl.req.eos) eos[i] = (int64_t)*sl.req.eos;
            sp[i] = fl_sampling{(double)sl.req.temperature, 0, sl.draws};
        }
        check(fl_batch_decode_each(batch_, first.data(), pos.data(), n, eos.data(), sp.data(), out.data(), n_out.data()), "Model forward pass failed");
        batch_steps += n;
        for (size_t i = 0; i < B; i++) {
            Slot &sl = slots_[i];
            if (!sl.active) continue;
            bool done = n_out[i] < n;                              // its EOS came up inside the chunk
            for (size_t k = 0; k < n_out[i]; k++) {
                if (sl.emitted >= sl.req.max_tokens) { done = true; break; }
                sl.emitted++;                                      // (tokens handed to the callback, the refused one included)
                if (!sl.req.on_token(out[i * n + k])) { done = true; break; }
            }
            if (!done && sl.emitted >= sl.req.max_tokens) done = true;
            sl.pos += n; sl.calls += n; sl.draws += n; sl.tok = out[i * n + n - 1];
            if (!done && sl.pos >= cap_) done = true;               // the slot's cache is full: the stream ends here
            if (done) finish(sl);
        }
        return true;
    }
    void run() { while (step()) {} }

  private:
    struct Request {
        uint64_t id = 0; std::vector<uint32_t> prompt; size_t max_tokens = 0; float temperature = 0.f; std::optional<uint32_t> eos;
        OnToken on_token; OnDone on_done;
    };
    struct Slot { fl_cache *cache = nullptr; bool active = false; Request req; size_t pos = 0, calls = 0, emitted = 0; uint32_t tok = 0; uint64_t draws = 0; };

    void admit(Slot &sl, Request r) {
        sl.req = std::move(r); sl.emitted = 0; sl.active = true;
        fl_cache_reset(sl.cache);
        const fl_sampling sp{(double)sl.req.temperature, 0, 0};
        uint32_t tok = 0;
        check(fl_forward_sample(model_->m, sl.cache, sl.req.prompt.data(), sl.req.prompt.size(), 0, &sp, &tok), "Model forward pass failed");
        prefills++;
        sl.pos = sl.req.prompt.size(); sl.calls = 1; sl.tok = tok; sl.draws = 1;
        if (sl.req.max_tokens == 0 || (sl.req.eos && tok == *sl.req.eos)) { finish(sl); return; }
        sl.emitted = 1;
        if (!sl.req.on_token(tok)) { finish(sl); return; }
        if (sl.emitted >= sl.req.max_tokens || sl.pos >= cap_) finish(sl);
    }
    void finish(Slot &sl) {
        sl.active = false;
        OnDone cb = std::move(sl.req.on_done);
        const size_t n = sl.emitted;
        sl.req = Request{};
        if (cb) cb(n);
    }
    void release() {
        if (batch_) { fl_batch_destroy(batch_); batch_ = nullptr; }
        for (auto &sl : slots_) if (sl.cache) { fl_cache_destroy(sl.cache); sl.cache = nullptr; }
    }

    std::shared_ptr<detail::ModelHandle> model_;
    size_t chunk_, cap_ = 0;
    bool counter_ = false;
    std::vector<Slot> slots_;
    std::deque<Request> queue_;
    fl_batch *batch_ = nullptr;
    uint64_t next_id_ = 0;
};

}  // namespace fastllm
