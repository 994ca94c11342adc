#include "json.h"

#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace kjarni {

namespace {

struct Parser {
    const char* p;
    const char* end;

    [[noreturn]] void fail(const char* msg) const { throw std::runtime_error(std::string("JSON parse error: ") + msg); }

    void skip_ws()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }

    static void append_utf8(std::string& out, uint32_t cp)
    {
        if (cp < 0x80) {
            out.push_back((char)cp);
        } else if (cp < 0x800) {
            out.push_back((char)(0xC0 | (cp >> 6)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            out.push_back((char)(0xE0 | (cp >> 12)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            out.push_back((char)(0xF0 | (cp >> 18)));
            out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }

    uint32_t hex4()
    {
        if (end - p < 4) fail("short \\u escape");
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("bad hex digit");
        }
        return v;
    }

    std::string parse_string()
    {
        if (p >= end || *p != '"') fail("expected string");
        ++p;
        std::string out;
        while (true) {
            if (p >= end) fail("unterminated string");
            char c = *p++;
            if (c == '"') break;
            if (c != '\\') {
                out.push_back(c);
                continue;
            }
            if (p >= end) fail("bad escape");
            char e = *p++;
            switch (e) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                uint32_t cp = hex4();
                if (cp >= 0xD800 && cp <= 0xDBFF && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
                    const char* save = p;
                    p += 2;
                    uint32_t lo = hex4();
                    if (lo >= 0xDC00 && lo <= 0xDFFF)
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    else
                        p = save;
                }
                append_utf8(out, cp);
                break;
            }
            default: fail("unknown escape");
            }
        }
        return out;
    }

    Json parse_value(int depth)
    {
        if (depth > 256) fail("nesting too deep");
        skip_ws();
        if (p >= end) fail("unexpected end");
        Json v;
        char c = *p;
        if (c == '{') {
            ++p;
            v.type = Json::Object;
            skip_ws();
            if (p < end && *p == '}') {
                ++p;
                return v;
            }
            while (true) {
                skip_ws();
                std::string key = parse_string();
                skip_ws();
                if (p >= end || *p != ':') fail("expected ':'");
                ++p;
                v.obj.emplace_back(std::move(key), parse_value(depth + 1));
                skip_ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == '}') {
                    ++p;
                    break;
                }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            ++p;
            v.type = Json::Array;
            skip_ws();
            if (p < end && *p == ']') {
                ++p;
                return v;
            }
            while (true) {
                v.arr.push_back(parse_value(depth + 1));
                skip_ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == ']') {
                    ++p;
                    break;
                }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.type = Json::String;
            v.str = parse_string();
        } else if (c == 't' && end - p >= 4 && std::memcmp(p, "true", 4) == 0) {
            p += 4;
            v.type = Json::Bool;
            v.b = true;
        } else if (c == 'f' && end - p >= 5 && std::memcmp(p, "false", 5) == 0) {
            p += 5;
            v.type = Json::Bool;
            v.b = false;
        } else if (c == 'n' && end - p >= 4 && std::memcmp(p, "null", 4) == 0) {
            p += 4;
            v.type = Json::Null;
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            const char* s = p;
            while (p < end && (*p == '-' || *p == '+' || *p == '.' || *p == 'e' || *p == 'E' ||
                               (*p >= '0' && *p <= '9')))
                ++p;
            // std::from_chars is locale-independent: a host that called setlocale(LC_NUMERIC, "de_DE") must still
            // read "1e-12" / "0.1" (strtod would stop at the '.', silently changing layer_norm_eps or rope_theta).
            const char* q = (*s == '+') ? s + 1 : s;
            const std::from_chars_result r = std::from_chars(q, p, v.num);
            if (r.ec != std::errc() || r.ptr != p) fail("bad number");
            v.type = Json::Number;
        } else if (c == 'N' && end - p >= 3 && std::memcmp(p, "NaN", 3) == 0) {
            p += 3;  // Python's json writes NaN/Infinity into some config files
            v.type = Json::Number;
            v.num = std::nan("");
        } else {
            fail("unexpected character");
        }
        return v;
    }
};

}  // namespace

Json Json::parse(const char* data, size_t len)
{
    Parser ps{data, data + len};
    Json v = ps.parse_value(0);
    ps.skip_ws();
    if (ps.p != ps.end) ps.fail("trailing characters");
    return v;
}

Json Json::parse(const std::string& text) { return parse(text.data(), text.size()); }

const Json* Json::find(const std::string& key) const
{
    if (type != Object) return nullptr;
    for (const auto& kv : obj)
        if (kv.first == key) return &kv.second;
    return nullptr;
}

const Json& Json::at(const std::string& key) const
{
    const Json* j = find(key);
    if (!j) throw std::runtime_error("JSON: missing key '" + key + "'");
    return *j;
}

int64_t Json::get_int(const std::string& key, int64_t dflt) const
{
    const Json* j = find(key);
    return (j && j->type == Number) ? (int64_t)j->num : dflt;
}

double Json::get_double(const std::string& key, double dflt) const
{
    const Json* j = find(key);
    return (j && j->type == Number) ? j->num : dflt;
}

bool Json::get_bool(const std::string& key, bool dflt) const
{
    const Json* j = find(key);
    return (j && j->type == Bool) ? j->b : dflt;
}

std::string Json::get_string(const std::string& key, const std::string& dflt) const
{
    const Json* j = find(key);
    return (j && j->type == String) ? j->str : dflt;
}

}  // namespace kjarni
