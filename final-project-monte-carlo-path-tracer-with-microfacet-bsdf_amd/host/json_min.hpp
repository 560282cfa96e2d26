// A small JSON reader for conf.json (the reference vendors nlohmann::json for this; only the accessors its main()
// uses are provided: operator[], contains, is_*, size, typed conversion with a type error on mismatch).
#pragma once
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

class Json {
  public:
    enum Type { Null, Bool, Number, String, Array, Object };
    Type type = Null;
    bool b = false;
    double num = 0;
    bool is_float = false;  // written with '.', 'e' or 'E' (nlohmann's is_number_float)
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;

    static Json parse(const std::string &text) {
        size_t i = 0;
        Json v = value(text, i);
        ws(text, i);
        if (i != text.size()) throw std::runtime_error("json: trailing characters");
        return v;
    }
    bool is_null() const { return type == Null; }
    bool is_boolean() const { return type == Bool; }
    bool is_number() const { return type == Number; }
    bool is_number_float() const { return type == Number && is_float; }
    bool is_string() const { return type == String; }
    bool is_array() const { return type == Array; }
    size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : (type == Null ? 0 : 1)); }
    bool contains(const std::string &k) const {
        for (const auto &kv : obj)
            if (kv.first == k) return true;
        return false;
    }
    const Json &operator[](const std::string &k) const {
        static const Json null_value;
        if (type != Object && type != Null) throw std::runtime_error("json: cannot use operator[] with a string key on a non-object");
        for (const auto &kv : obj)
            if (kv.first == k) return kv.second;
        return null_value;
    }
    const Json &operator[](size_t k) const {
        if (type != Array || k >= arr.size()) throw std::runtime_error("json: array index out of range");
        return arr[k];
    }
    float as_float() const {
        if (type != Number) throw std::runtime_error("json: type must be number");
        return (float)num;
    }
    int as_int() const {
        if (type != Number) throw std::runtime_error("json: type must be number");
        return (int)num;
    }
    bool as_bool() const {
        if (type != Bool) throw std::runtime_error("json: type must be boolean");
        return b;
    }
    const std::string &as_string() const {
        if (type != String) throw std::runtime_error("json: type must be string");
        return str;
    }

  private:
    static void ws(const std::string &t, size_t &i) {
        while (i < t.size() && (t[i] == ' ' || t[i] == '\t' || t[i] == '\n' || t[i] == '\r')) ++i;
    }
    static Json value(const std::string &t, size_t &i) {
        ws(t, i);
        if (i >= t.size()) throw std::runtime_error("json: unexpected end of input");
        Json v;
        const char c = t[i];
        if (c == '{') {
            v.type = Object;
            ++i;
            ws(t, i);
            if (i < t.size() && t[i] == '}') { ++i; return v; }
            while (true) {
                ws(t, i);
                Json k = value(t, i);
                if (k.type != String) throw std::runtime_error("json: object key must be a string");
                ws(t, i);
                if (i >= t.size() || t[i] != ':') throw std::runtime_error("json: expected ':'");
                ++i;
                v.obj.emplace_back(k.str, value(t, i));
                ws(t, i);
                if (i < t.size() && t[i] == ',') { ++i; continue; }
                if (i < t.size() && t[i] == '}') { ++i; return v; }
                throw std::runtime_error("json: expected ',' or '}'");
            }
        }
        if (c == '[') {
            v.type = Array;
            ++i;
            ws(t, i);
            if (i < t.size() && t[i] == ']') { ++i; return v; }
            while (true) {
                v.arr.push_back(value(t, i));
                ws(t, i);
                if (i < t.size() && t[i] == ',') { ++i; continue; }
                if (i < t.size() && t[i] == ']') { ++i; return v; }
                throw std::runtime_error("json: expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.type = String;
            ++i;
            while (i < t.size() && t[i] != '"') {
                if (t[i] == '\\' && i + 1 < t.size()) {
                    const char e = t[++i];
                    v.str += (e == 'n') ? '\n' : (e == 't') ? '\t' : e;
                } else {
                    v.str += t[i];
                }
                ++i;
            }
            if (i >= t.size()) throw std::runtime_error("json: unterminated string");
            ++i;
            return v;
        }
        if (t.compare(i, 4, "true") == 0) { v.type = Bool; v.b = true; i += 4; return v; }
        if (t.compare(i, 5, "false") == 0) { v.type = Bool; v.b = false; i += 5; return v; }
        if (t.compare(i, 4, "null") == 0) { i += 4; return v; }
        const size_t s = i;
        while (i < t.size() && (std::string("+-0123456789.eE").find(t[i]) != std::string::npos)) ++i;
        if (s == i) throw std::runtime_error("json: unexpected character");
        const std::string tok = t.substr(s, i - s);
        v.type = Number;
        v.num = std::strtod(tok.c_str(), nullptr);
        v.is_float = tok.find_first_of(".eE") != std::string::npos;
        return v;
    }
};
