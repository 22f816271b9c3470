// Sanitizer driver for the host layer's input parsers (safetensors reader, index file, config.json): built by
// tests/test_host_sanitizers.py with -fsanitize=address,undefined and fed a corpus of truncated / oversized / overflowing /
// hostile files.  Every input must end in "accepted" or in a fastllm::Error ("rejected"); anything the sanitizers see --
// an out-of-bounds read, a signed overflow, a float -> integer cast out of range -- aborts the run.  CPU only: nothing here
// touches the GPU library (the parsers are header-only).
//
//   fuzz_host_inputs <path>...      *.safetensors -> SafetensorsFile;  a directory -> Checkpoint (index + shards);
//                                   *.json -> Json::parse, architecture_of, BaseModelConfig::from_json + validation
#include <sys/stat.h>

#include <cstdio>
#include <string>

#include "../../fastllm_amd/host/safetensors.hpp"

using namespace fastllm;

static bool ends_with(const std::string &s, const char *suf) {
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// touch every byte a reader would hand to fl_model_create: a shape larger than its bytes shows up here as a read past the mapping
static uint64_t checksum(const Tensor &t) {
    size_t esz = t.dtype == DType::F32 ? 4 : 2;
    uint64_t n = esz, s = 0;
    for (auto d : t.shape) n *= (uint64_t)d;
    const unsigned char *p = static_cast<const unsigned char *>(t.data);
    for (uint64_t i = 0; i < n; i++) s += p[i];
    return s;
}

template <class F>
static void attempt(const std::string &what, const std::string &path, F &&f) {
    try {
        const std::string r = f();
        std::printf("accepted %s %s %s\n", what.c_str(), path.c_str(), r.c_str());
    } catch (const Error &e) {
        std::printf("rejected %s %s (%d) %s\n", what.c_str(), path.c_str(), e.code, e.what());
    } catch (const Panic &e) {
        std::printf("rejected %s %s (panic) %s\n", what.c_str(), path.c_str(), e.what());
    }
}

int main(int argc, char **argv) {
    for (int a = 1; a < argc; a++) {
        const std::string path = argv[a];
        struct stat st;
        if (::stat(path.c_str(), &st) != 0) { std::printf("missing %s\n", path.c_str()); continue; }
        if (S_ISDIR(st.st_mode)) {
            attempt("checkpoint", path, [&] {
                Checkpoint ck(path);
                uint64_t s = 0;
                for (auto &kv : ck.tensors) s += checksum(kv.second);
                return std::to_string(ck.tensors.size()) + " tensors, byte sum " + std::to_string(s);
            });
        } else if (ends_with(path, ".safetensors")) {
            attempt("safetensors", path, [&] {
                SafetensorsFile f(path);
                uint64_t s = 0;
                for (auto &kv : f.tensors()) s += checksum(kv.second);
                return std::to_string(f.tensors().size()) + " tensors, byte sum " + std::to_string(s);
            });
        } else {
            const std::string text = read_text(path);
            attempt("json", path, [&] { Json j = Json::parse(text); return std::string("kind ") + std::to_string((int)j.kind); });
            attempt("architecture", path, [&] { return architecture_of(text); });
            attempt("config", path, [&] {
                BaseModelConfig c = BaseModelConfig::from_json(text);
                const size_t d = c.validate_head_dimensions();
                c.validate_gqa_config();
                const fl_config f = c.to_fl(FL_FAMILY_LLAMA, false);
                return "head_dim " + std::to_string(d) + " hidden " + std::to_string((long long)f.hidden_size);
            });
        }
    }
    std::printf("fuzz driver done\n");
    return 0;
}
