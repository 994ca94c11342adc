// Minimal JSON reader for config.json, tokenizer.json and the safetensors header.
// Host-side plumbing only (the reference uses serde_json for the same files:
// crates/kjarni-transformers/src/weights/model_weights.rs:45-282).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace kjarni {

class Json {
public:
    enum Type { Null, Bool, Number, String, Array, Object };

    Type type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    // Insertion order is kept (vocab files rely on it only through explicit ids,
    // but error messages and iteration stay deterministic).
    std::vector<std::pair<std::string, Json>> obj;

    static Json parse(const std::string& text);
    static Json parse(const char* data, size_t len);

    bool is_null() const { return type == Null; }
    bool is_object() const { return type == Object; }
    bool is_array() const { return type == Array; }
    bool is_string() const { return type == String; }
    bool is_number() const { return type == Number; }
    bool is_bool() const { return type == Bool; }

    // Object lookup; returns nullptr when absent or not an object.
    const Json* find(const std::string& key) const;
    const Json& at(const std::string& key) const;  // throws when absent

    int64_t as_int(int64_t dflt = 0) const { return type == Number ? (int64_t)num : dflt; }
    double as_double(double dflt = 0.0) const { return type == Number ? num : dflt; }
    bool as_bool(bool dflt = false) const { return type == Bool ? b : dflt; }
    const std::string& as_string() const { return str; }

    int64_t get_int(const std::string& key, int64_t dflt) const;
    double get_double(const std::string& key, double dflt) const;
    bool get_bool(const std::string& key, bool dflt) const;
    std::string get_string(const std::string& key, const std::string& dflt) const;
};

}  // namespace kjarni
