// Minimal JSON reader for the glTF front-end (the reference used rapidjson, whose submodule is
// absent).  Recursive descent, UTF-8 pass-through, numbers kept as double via
// strtod (correctly rounded; the reference's GetFloat() then narrows to float, and every number
// a glTF exporter writes is the decimal expansion of a float, so both readers yield the same float).
#pragma once
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rtamd {

class Json {
public:
    enum Type { Null, Bool, Number, String, Array, Object };
    Type type = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj; // insertion order preserved

    bool has(const char *key) const { return find(key) != nullptr; }
    const Json *find(const char *key) const {
        if (type != Object) return nullptr;
        for (auto &kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    const Json &at(const char *key) const {
        const Json *j = find(key);
        if (!j) throw std::runtime_error(std::string("glTF: missing key '") + key + "'");
        return *j;
    }
    const Json &idx(size_t i) const {
        if (type != Array || i >= arr.size()) throw std::runtime_error("glTF: array index out of range");
        return arr[i];
    }
    size_t size() const { return type == Array ? arr.size() : obj.size(); }
    float as_float() const { need(Number); return (float)num; }
    double as_double() const { need(Number); return num; }
    uint32_t as_uint() const { need(Number); return (uint32_t)num; }
    const std::string &as_string() const { need(String); return str; }

    static Json parse(const std::string &text) {
        Parser p{text.c_str(), text.c_str() + text.size()};
        p.ws();
        Json j = p.value();
        p.ws();
        if (p.cur != p.end) p.fail("trailing characters");
        return j;
    }

private:
    void need(Type t) const {
        if (type != t) throw std::runtime_error("glTF: JSON value has unexpected type");
    }
    struct Parser {
        const char *cur, *end;
        [[noreturn]] void fail(const char *msg) { throw std::runtime_error(std::string("JSON parse error: ") + msg); }
        void ws() {
            while (cur < end && (*cur == ' ' || *cur == '\t' || *cur == '\n' || *cur == '\r')) cur++;
        }
        Json value() {
            if (cur >= end) fail("unexpected end");
            switch (*cur) {
            case '{': return object();
            case '[': return array();
            case '"': { Json j; j.type = String; j.str = string(); return j; }
            case 't': lit("true"); { Json j; j.type = Bool; j.b = true; return j; }
            case 'f': lit("false"); { Json j; j.type = Bool; j.b = false; return j; }
            case 'n': lit("null"); return Json();
            default: return number();
            }
        }
        void lit(const char *s) {
            size_t n = strlen(s);
            if ((size_t)(end - cur) < n || strncmp(cur, s, n) != 0) fail("bad literal");
            cur += n;
        }
        Json number() {
            char *e = nullptr;
            double d = strtod(cur, &e);
            if (e == cur) fail("bad number");
            cur = e;
            Json j; j.type = Number; j.num = d;
            return j;
        }
        std::string string() {
            std::string out;
            cur++; // opening quote
            while (cur < end && *cur != '"') {
                if (*cur == '\\') {
                    cur++;
                    if (cur >= end) fail("bad escape");
                    switch (*cur) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': { // keep BMP code points as UTF-8
                        if (end - cur < 5) fail("bad \\u escape");
                        unsigned cp = (unsigned)strtoul(std::string(cur + 1, cur + 5).c_str(), nullptr, 16);
                        cur += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += *cur; break; // \" \\ \/
                    }
                    cur++;
                } else out += *cur++;
            }
            if (cur >= end) fail("unterminated string");
            cur++;
            return out;
        }
        Json array() {
            Json j; j.type = Array;
            cur++; ws();
            if (cur < end && *cur == ']') { cur++; return j; }
            for (;;) {
                ws();
                j.arr.push_back(value());
                ws();
                if (cur >= end) fail("unterminated array");
                if (*cur == ',') { cur++; continue; }
                if (*cur == ']') { cur++; return j; }
                fail("expected , or ]");
            }
        }
        Json object() {
            Json j; j.type = Object;
            cur++; ws();
            if (cur < end && *cur == '}') { cur++; return j; }
            for (;;) {
                ws();
                if (cur >= end || *cur != '"') fail("expected key");
                std::string k = string();
                ws();
                if (cur >= end || *cur != ':') fail("expected :");
                cur++; ws();
                j.obj.emplace_back(std::move(k), value());
                ws();
                if (cur >= end) fail("unterminated object");
                if (*cur == ',') { cur++; continue; }
                if (*cur == '}') { cur++; return j; }
                fail("expected , or }");
            }
        }
    };
};

} // namespace rtamd
