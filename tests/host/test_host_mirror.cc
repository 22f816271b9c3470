// C++ unit tests of the host mirror; each case restates one of the reference's own unit tests
// (file:line given) against the mirrored types.  No GPU needed: nothing here calls forward.
#include <cassert>
#include <cstdio>

#include "../../fastllm_amd/host/fastllm_host.hpp"

using namespace fastllm;

#define EXPECT(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)
template <class E, class F> bool throws(F &&f) { try { f(); } catch (const E &) { return true; } catch (...) { return false; } return false; }

static const char *kCfg = R"({"architectures":["MistralForCausalLM"],"hidden_size":4096,"intermediate_size":14336,
 "vocab_size":32000,"num_hidden_layers":32,"num_attention_heads":32,"num_key_value_heads":8,"rms_norm_eps":1e-05,
 "rope_theta":10000.0,"max_position_embeddings":32768,"sliding_window":4096,"torch_dtype":"bfloat16","rope_scaling":null,
 "nested":{"a":[1,2,{"b":"}"}]}})";

static int test_common_cache_operations() {            // cache.rs:49-79
    CommonCache c;
    EXPECT(c.get_offset() == 0);
    c.increment_offset(); EXPECT(c.get_offset() == 1);
    c.increment_offset(); EXPECT(c.get_offset() == 2);
    c.reset(); EXPECT(c.get_offset() == 0);
    ModelCache *dyn = &c; dyn->increment_offset();     // as_any_mut / trait-object use (cache.rs:81-89)
    EXPECT(c.get_offset() == 1);
    return 0;
}
static int test_family_cache_operations() {            // llama.rs:168-205, mistral.rs:255-271, qwen.rs:192-208
    LlamaCache l; MistralCache m; QwenCache q;
    for (ModelCache *c : {(ModelCache *)&l, (ModelCache *)&m, (ModelCache *)&q}) {
        EXPECT(c->get_offset() == 0);
        c->increment_offset(); EXPECT(c->get_offset() == 1);
        c->reset(); EXPECT(c->get_offset() == 0);
    }
    return 0;
}
static int test_config_conversion() {                  // llama.rs:208-232, mistral.rs:274-300, qwen.rs:211-237
    BaseModelConfig c = BaseModelConfig::from_json(kCfg);
    EXPECT(c.hidden_size == 4096 && c.intermediate_size == 14336 && c.vocab_size == 32000);
    EXPECT(c.num_hidden_layers == 32 && c.num_attention_heads == 32 && c.num_key_value_heads.value() == 8);
    EXPECT(c.rms_norm_eps == 1e-5 && c.rope_theta.value() == 10000.0 && c.sliding_window.value() == 4096);
    EXPECT(c.torch_dtype.value() == "bfloat16");
    fl_config f = c.to_fl(FL_FAMILY_MISTRAL, false);
    EXPECT(f.hidden_size == 4096 && f.num_key_value_heads == 8 && f.sliding_window == 4096 && f.family == FL_FAMILY_MISTRAL);
    // optional fields absent -> 0 -> the library applies the reference defaults
    BaseModelConfig d = BaseModelConfig::from_json(R"({"hidden_size":64,"intermediate_size":128,"vocab_size":10,
        "num_hidden_layers":1,"num_attention_heads":4,"rms_norm_eps":1e-6})");
    EXPECT(!d.num_key_value_heads && !d.rope_theta && !d.max_position_embeddings && !d.sliding_window);
    EXPECT(d.to_fl(FL_FAMILY_LLAMA, false).num_key_value_heads == 0);
    // serde: a missing mandatory field is an error
    EXPECT(throws<Error>([] { BaseModelConfig::from_json(R"({"hidden_size":64})"); }));
    return 0;
}
static int test_head_dim_and_gqa_validation() {        // config.rs:61-144, mistral.rs:347-363
    BaseModelConfig c = BaseModelConfig::from_json(kCfg);
    EXPECT(c.validate_head_dimensions() == 128);
    c.validate_gqa_config();
    BaseModelConfig bad = c; bad.hidden_size = 4097;
    EXPECT(throws<Error>([&] { bad.validate_head_dimensions(); }));
    bad = c; bad.hidden_size = 96; bad.num_attention_heads = 32;       // head_dim 3: odd
    EXPECT(throws<Error>([&] { bad.validate_head_dimensions(); }));
    bad = c; bad.num_key_value_heads = 5;
    EXPECT(throws<Error>([&] { bad.validate_gqa_config(); }));
    EXPECT(MistralWithConfig::get_head_dim(4096, 32) == 128);
    EXPECT(throws<Panic>([] { MistralWithConfig::get_head_dim(4097, 32); }));     // #[should_panic] mistral.rs:356-363
    EXPECT(throws<Panic>([&] { BaseModelConfig b = BaseModelConfig::from_json(kCfg); b.num_key_value_heads = 5; MistralWithConfig::validate(b); }));
    return 0;
}
static int test_architecture_support() {               // model_registry.rs:225-277, llama.rs:157, mistral.rs:244, qwen.rs:178
    EXPECT(LlamaWithConfig::supports_architecture("LlamaForCausalLM"));
    EXPECT(!LlamaWithConfig::supports_architecture("MistralForCausalLM"));
    EXPECT(MistralWithConfig::supports_architecture("MistralForCausalLM"));
    EXPECT(QwenWithConfig::supports_architecture("Qwen2ForCausalLM"));
    EXPECT(QwenWithConfig::supports_architecture("Qwen2_5_VLForConditionalGeneration"));
    EXPECT(!QwenWithConfig::supports_architecture("Qwen3ForCausalLM"));
    EXPECT(std::string(LlamaWithConfig::get_family()) == "Llama" && std::string(MistralWithConfig::get_family()) == "Mistral" &&
           std::string(QwenWithConfig::get_family()) == "Qwen");
    return 0;
}
static int test_no_cpu_path() {
    // initialize_model on Device::Cpu must fail loudly: the product has no CPU fallback
    BaseModelConfig c = BaseModelConfig::from_json(kCfg);
    TensorMap empty;
    EXPECT(throws<Error>([&] { LlamaWithConfig::initialize_model(c, empty, DType::BF16, Device::cpu()); }));
    EXPECT(throws<Error>([&] { LlamaWithConfig::initialize_cache(Device::cpu(), DType::BF16); }));   // no model yet
    return 0;
}
static int test_argmax_last_max_wins() {               // LogitsProcessor ArgMax, Rust max_by semantics
    float v[6] = {0.f, 3.f, 1.f, 3.f, -1.f, 3.f};
    EXPECT(argmax_last(v, 6) == 5);
    float w[3] = {2.f, 1.f, 0.f};
    EXPECT(argmax_last(w, 3) == 0);
    return 0;
}

int main() {
    int rc = 0;
    rc |= test_common_cache_operations();
    rc |= test_family_cache_operations();
    rc |= test_config_conversion();
    rc |= test_head_dim_and_gqa_validation();
    rc |= test_architecture_support();
    rc |= test_no_cpu_path();
    rc |= test_argmax_last_max_wins();
    std::printf(rc ? "HOST MIRROR TESTS FAILED\n" : "host mirror tests ok\n");
    return rc;
}
