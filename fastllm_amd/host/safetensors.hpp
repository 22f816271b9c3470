// safetensors.hpp -- "next" row N1 of the hot-path scope table: the data format on the input side of
// initialize_model.  The reference reads model.safetensors (or the shards named by
// model.safetensors.index.json -> weight_map) with candle_core::safetensors::load_buffer and hands the
// resulting HashMap<String, Tensor> to M::initialize_model
// (/root/reference/src/providers/huggingface/huggingface.rs:83-135).
//
// Format (safetensors 0.x): 8-byte little-endian header length N, N bytes of JSON
//   { "<name>": {"dtype": "BF16"|"F16"|"F32"|..., "shape": [..], "data_offsets": [begin, end]}, ...,
//     "__metadata__": {...} },
// then the raw little-endian tensor bytes; offsets are relative to the end of the header.
// The file is mmap'ed and the tensors are views into the mapping (the reference reads the whole file
// into RAM first, quirk C.8); fl_model_create copies them to HBM.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <fstream>
#include <map>
#include <set>
#include <sstream>

#include "fastllm_host.hpp"

namespace fastllm {

// ------------------------------------------------------------------------------------ tiny JSON DOM
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;       // insertion order kept

    const Json *get(const std::string &k) const {
        for (auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    static constexpr int kMaxDepth = 64;                  // checkpoint headers nest two levels; a bound keeps hostile input off the stack
    static Json parse(const std::string &s) {
        size_t i = 0;
        Json j = value(s, i, 0);
        ws(s, i);
        if (i != s.size()) throw Error(FL_ERR_BAD_ARGUMENT, "trailing characters in JSON");
        return j;
    }

   private:
    static void ws(const std::string &s, size_t &i) { while (i < s.size() && std::isspace((unsigned char)s[i])) i++; }
    static Json value(const std::string &s, size_t &i, int depth) {
        ws(s, i);
        if (i >= s.size()) throw Error(FL_ERR_BAD_ARGUMENT, "unexpected end of JSON");
        if (depth > kMaxDepth) throw Error(FL_ERR_BAD_ARGUMENT, "JSON nested too deeply");
        Json j;
        char c = s[i];
        if (c == '{') {
            j.kind = Obj; i++; ws(s, i);
            if (i < s.size() && s[i] == '}') { i++; return j; }
            while (true) {
                ws(s, i);
                Json k = value(s, i, depth + 1);
                if (k.kind != Str) throw Error(FL_ERR_BAD_ARGUMENT, "JSON object key must be a string");
                ws(s, i);
                if (i >= s.size() || s[i] != ':') throw Error(FL_ERR_BAD_ARGUMENT, "JSON: expected ':'");
                i++;
                j.obj.emplace_back(k.str, value(s, i, depth + 1));
                ws(s, i);
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == '}') { i++; return j; }
                throw Error(FL_ERR_BAD_ARGUMENT, "JSON: expected ',' or '}'");
            }
        }
        if (c == '[') {
            j.kind = Arr; i++; ws(s, i);
            if (i < s.size() && s[i] == ']') { i++; return j; }
            while (true) {
                j.arr.push_back(value(s, i, depth + 1));
                ws(s, i);
                if (i < s.size() && s[i] == ',') { i++; continue; }
                if (i < s.size() && s[i] == ']') { i++; return j; }
                throw Error(FL_ERR_BAD_ARGUMENT, "JSON: expected ',' or ']'");
            }
        }
        if (c == '"') {
            j.kind = Str; i++;
            while (i < s.size() && s[i] != '"') {
                if (s[i] == '\\' && i + 1 < s.size()) {
                    char e = s[++i];
                    switch (e) {
                        case 'n': j.str += '\n'; break; case 't': j.str += '\t'; break; case 'r': j.str += '\r'; break;
                        case 'b': j.str += '\b'; break; case 'f': j.str += '\f'; break;
                        case 'u':                                          // names in checkpoints are ASCII
                            if (i + 4 >= s.size()) throw Error(FL_ERR_BAD_ARGUMENT, "JSON: truncated \\u escape");
                            j.str += '?'; i += 4; break;
                        default: j.str += e;
                    }
                    i++;
                } else j.str += s[i++];
            }
            if (i >= s.size()) throw Error(FL_ERR_BAD_ARGUMENT, "JSON: unterminated string");
            i++;
            return j;
        }
        if (!s.compare(i, 4, "true")) { j.kind = Bool; j.b = true; i += 4; return j; }
        if (!s.compare(i, 5, "false")) { j.kind = Bool; i += 5; return j; }
        if (!s.compare(i, 4, "null")) { i += 4; return j; }
        char *end = nullptr;
        j.num = std::strtod(s.c_str() + i, &end);
        if (end == s.c_str() + i) throw Error(FL_ERR_BAD_ARGUMENT, "JSON: unexpected character");
        j.kind = Num; i = (size_t)(end - s.c_str());
        return j;
    }
};

// a JSON number that must be a non-negative integer below 2^53 (shape entries, data offsets): what serde gives the
// safetensors crate as usize; anything else -- negative, fractional, NaN, 1e300 -- is refused, never cast
inline bool json_index(const Json &j, uint64_t *out) {
    if (j.kind != Json::Num || !(j.num >= 0.0) || !(j.num <= 9007199254740992.0)) return false;
    const uint64_t v = (uint64_t)j.num;
    if ((double)v != j.num) return false;
    *out = v;
    return true;
}

// ------------------------------------------------------------------------------------ one file
// Validation follows the safetensors crate's (what candle_core::safetensors::load_buffer runs, huggingface.rs:88,125): header
// length bounded, every tensor's byte range equal to shape x dtype size (checked multiplication), ranges inside the data
// section, non-overlapping and without holes, and the last one ending at the end of the file.
class SafetensorsFile {
   public:
    explicit SafetensorsFile(const std::string &path) : path_(path) {
        try { load(path); } catch (...) { release(); throw; }      // (a throwing constructor's destructor never runs)
    }
    ~SafetensorsFile() { release(); }
    SafetensorsFile(const SafetensorsFile &) = delete;
    const std::map<std::string, Tensor> &tensors() const { return tensors_; }
    static constexpr uint64_t kMaxHeader = 100000000;        // the safetensors crate's MAX_HEADER_SIZE

   private:
    void release() {
        if (map_ && map_ != MAP_FAILED) ::munmap(map_, size_);
        if (fd_ >= 0) ::close(fd_);
        map_ = nullptr; fd_ = -1;
    }
    void load(const std::string &path) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw Error(FL_ERR_BAD_ARGUMENT, "Failed to read " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0 || st.st_size < 8) throw Error(FL_ERR_BAD_ARGUMENT, "Failed to load tensors from safetensors: " + path + " is too short");
        size_ = (size_t)st.st_size;
        map_ = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (map_ == MAP_FAILED) { map_ = nullptr; throw Error(FL_ERR_OOM, "mmap failed for " + path); }
        const unsigned char *p = static_cast<const unsigned char *>(map_);
        uint64_t n = 0;
        for (int i = 7; i >= 0; i--) n = (n << 8) | p[i];                     // little-endian u64
        if (n > size_ - 8) fail("header length exceeds the file");
        if (n > kMaxHeader) fail("header too large");
        Json hdr = Json::parse(std::string(reinterpret_cast<const char *>(p + 8), (size_t)n));
        if (hdr.kind != Json::Obj) fail("header is not a JSON object");
        const size_t data0 = 8 + (size_t)n, data_len = size_ - data0;
        std::vector<std::pair<uint64_t, uint64_t>> spans;
        for (auto &kv : hdr.obj) {
            if (kv.first == "__metadata__") continue;
            const Json *dt = kv.second.get("dtype"), *sh = kv.second.get("shape"), *off = kv.second.get("data_offsets");
            if (!dt || dt->kind != Json::Str || !sh || sh->kind != Json::Arr || !off || off->kind != Json::Arr || off->arr.size() != 2)
                fail("malformed entry for tensor " + kv.first);
            Tensor t;
            size_t esz = 0;
            if (dt->str == "BF16") { t.dtype = DType::BF16; esz = 2; }
            else if (dt->str == "F16") { t.dtype = DType::F16; esz = 2; }
            else if (dt->str == "F32") { t.dtype = DType::F32; esz = 4; }
            else fail("unsupported dtype " + dt->str + " for tensor " + kv.first);
            if (sh->arr.size() > 8) fail("tensor " + kv.first + " has too many dimensions");
            uint64_t bytes = esz;
            for (auto &d : sh->arr) {
                uint64_t v = 0;
                if (!json_index(d, &v)) fail("shape of " + kv.first + " is not a list of non-negative integers");
                if (__builtin_mul_overflow(bytes, v, &bytes)) fail("shape of " + kv.first + " overflows");
                t.shape.push_back((int64_t)v);
            }
            uint64_t b = 0, e = 0;
            if (!json_index(off->arr[0], &b) || !json_index(off->arr[1], &e)) fail("data_offsets of " + kv.first + " are not non-negative integers");
            if (e < b || e > data_len || e - b != bytes) fail("data_offsets of " + kv.first + " do not match its shape");
            t.data = p + data0 + b;
            t.device = -1;
            if (!tensors_.emplace(kv.first, t).second) fail("tensor " + kv.first + " appears twice");
            spans.emplace_back(b, e);
        }
        std::sort(spans.begin(), spans.end());
        uint64_t at = 0;
        for (auto &sp : spans) {
            if (sp.first != at) fail(sp.first < at ? "tensor byte ranges overlap" : "tensor byte ranges leave a hole");
            at = sp.second;
        }
        if (at != data_len) fail("tensor data does not cover the file");
    }
    [[noreturn]] void fail(const std::string &m) const { throw Error(FL_ERR_BAD_ARGUMENT, "Failed to load tensors from safetensors (" + path_ + "): " + m); }
    std::string path_;
    int fd_ = -1;
    void *map_ = nullptr;
    size_t size_ = 0;
    std::map<std::string, Tensor> tensors_;
};

inline std::string read_text(const std::string &path) {
    std::ifstream f(path);
    if (!f) throw Error(FL_ERR_BAD_ARGUMENT, "Failed to read " + path);
    std::stringstream ss; ss << f.rdbuf();
    return ss.str();
}
inline bool file_exists(const std::string &p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }

// ------------------------------------------------------------------------------------ a checkpoint directory
// huggingface.rs:83-130: model.safetensors if present, else every file named in
// model.safetensors.index.json's weight_map; all tensors merged into one map.
struct Checkpoint {
    std::vector<std::unique_ptr<SafetensorsFile>> files;      // keep the mappings alive
    TensorMap tensors;
    explicit Checkpoint(const std::string &dir) {
        const std::string single = dir + "/model.safetensors";
        std::vector<std::string> names;
        if (file_exists(single)) names.push_back("model.safetensors");
        else {
            const std::string idx = dir + "/model.safetensors.index.json";
            if (!file_exists(idx)) throw Error(FL_ERR_BAD_ARGUMENT, "Failed to find either model.safetensors or model.safetensors.index.json");
            Json j = Json::parse(read_text(idx));
            const Json *wm = j.get("weight_map");
            if (!wm || wm->kind != Json::Obj) throw Error(FL_ERR_BAD_ARGUMENT, "Invalid index file format: missing or invalid weight_map");
            std::set<std::string> uniq;
            for (auto &kv : wm->obj) {
                if (kv.second.kind != Json::Str) continue;
                // a shard is a file of THIS directory (the reference fetches each name from the same hub repository)
                const std::string &fn = kv.second.str;
                if (fn.empty() || fn[0] == '/' || fn.find("..") != std::string::npos)
                    throw Error(FL_ERR_BAD_ARGUMENT, "Invalid index file format: shard name " + fn + " leaves the checkpoint directory");
                uniq.insert(fn);
            }
            names.assign(uniq.begin(), uniq.end());
        }
        for (auto &n : names) {
            files.emplace_back(new SafetensorsFile(dir + "/" + n));
            for (auto &kv : files.back()->tensors()) tensors[kv.first] = kv.second;
        }
    }
};

// model_registry.rs:129-152 + 169-182: architectures[0] of config.json, mapped by substring
inline std::string architecture_of(const std::string &config_json) {
    Json j = Json::parse(config_json);
    const Json *a = j.get("architectures");
    if (!a || a->kind != Json::Arr || a->arr.empty() || a->arr[0].kind != Json::Str)
        throw Error(FL_ERR_BAD_CONFIG, "No architecture found in config");
    return a->arr[0].str;
}
inline const char *get_family_from_architecture(const std::string &arch) {
    if (arch.find("Llama") != std::string::npos) return "Llama";
    if (arch.find("Mistral") != std::string::npos) return "Mistral";
    if (arch.find("Qwen") != std::string::npos) return "Qwen";
    return nullptr;
}

// load_model::<M> (huggingface.rs:18-139) without the Hub download and the tokenizer: a local directory
// holding config.json + safetensors.
template <class M>
Model<M> load_model(const std::string &dir, DType dtype, const Device &device) {
    const std::string cfg_text = read_text(dir + "/config.json");
    const std::string arch = architecture_of(cfg_text);
    if (!M::supports_architecture(arch))                                        // huggingface.rs:69-76
        throw Error(FL_ERR_BAD_CONFIG, "Model architecture mismatch: expected " + std::string(M::get_family()) + ", got " + arch);
    typename M::Config cfg = M::Config::from_json(cfg_text);
    Checkpoint ck(dir);
    auto r = M::initialize_model(cfg, ck.tensors, dtype, device);
    return Model<M>(r.first, device, r.second);
}

}  // namespace fastllm
