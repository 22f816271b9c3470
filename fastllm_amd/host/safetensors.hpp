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
    static Json parse(const std::string &s) {
        size_t i = 0;
        Json j = value(s, i);
        ws(s, i);
        if (i != s.size()) throw Error(FL_ERR_BAD_ARGUMENT, "trailing characters in JSON");
        return j;
    }

   private:
    static void ws(const std::string &s, size_t &i) { while (i < s.size() && std::isspace((unsigned char)s[i])) i++; }
    static Json value(const std::string &s, size_t &i) {
        ws(s, i);
        if (i >= s.size()) throw Error(FL_ERR_BAD_ARGUMENT, "unexpected end of JSON");
        Json j;
        char c = s[i];
        if (c == '{') {
            j.kind = Obj; i++; ws(s, i);
            if (i < s.size() && s[i] == '}') { i++; return j; }
            while (true) {
                ws(s, i);
                Json k = value(s, i);
                if (k.kind != Str) throw Error(FL_ERR_BAD_ARGUMENT, "JSON object key must be a string");
                ws(s, i);
                if (i >= s.size() || s[i] != ':') throw Error(FL_ERR_BAD_ARGUMENT, "JSON: expected ':'");
                i++;
                j.obj.emplace_back(k.str, value(s, i));
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
                j.arr.push_back(value(s, i));
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
                        case 'u': j.str += '?'; i += 4; break;            // names in checkpoints are ASCII
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

// ------------------------------------------------------------------------------------ one file
class SafetensorsFile {
   public:
    explicit SafetensorsFile(const std::string &path) : path_(path) {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw Error(FL_ERR_BAD_ARGUMENT, "Failed to read " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0 || st.st_size < 8) { ::close(fd_); throw Error(FL_ERR_BAD_ARGUMENT, "Failed to load tensors from safetensors: " + path + " is too short"); }
        size_ = (size_t)st.st_size;
        map_ = ::mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (map_ == MAP_FAILED) { ::close(fd_); throw Error(FL_ERR_OOM, "mmap failed for " + path); }
        const unsigned char *p = static_cast<const unsigned char *>(map_);
        uint64_t n = 0;
        for (int i = 7; i >= 0; i--) n = (n << 8) | p[i];                     // little-endian u64
        if (n > size_ - 8) fail("header length exceeds the file");
        Json hdr = Json::parse(std::string(reinterpret_cast<const char *>(p + 8), (size_t)n));
        if (hdr.kind != Json::Obj) fail("header is not a JSON object");
        const size_t data0 = 8 + (size_t)n, data_len = size_ - data0;
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
            size_t cnt = 1;
            for (auto &d : sh->arr) { t.shape.push_back((int64_t)d.num); cnt *= (size_t)d.num; }
            const size_t b = (size_t)off->arr[0].num, e = (size_t)off->arr[1].num;
            if (e < b || e > data_len || e - b != cnt * esz) fail("data_offsets of " + kv.first + " do not match its shape");
            t.data = p + data0 + b;
            t.device = -1;
            tensors_[kv.first] = t;
        }
    }
    ~SafetensorsFile() {
        if (map_ && map_ != MAP_FAILED) ::munmap(map_, size_);
        if (fd_ >= 0) ::close(fd_);
    }
    SafetensorsFile(const SafetensorsFile &) = delete;
    const std::map<std::string, Tensor> &tensors() const { return tensors_; }

   private:
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
            for (auto &kv : wm->obj) if (kv.second.kind == Json::Str) uniq.insert(kv.second.str);
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
