// json_min.hpp -- a small recursive-descent JSON reader, just enough for the two one-line
// configuration files of the path (parameter file: fwi_utils.py:46-83, survey file:
// fwi_utils.py:87-124 of the reference).  Replaces the vendored rapidjson of the reference
// (Src/rapidjson/, used by Src/Parameter.cpp and Src/Src_Rec.cu) for this purpose.
#pragma once
#include <cctype>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace sepfwi {

struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    bool is_int = false;  // number token had no '.', 'e' or 'E'
    std::string str;
    std::vector<JsonValue> arr;
    std::map<std::string, JsonValue> obj;

    bool has(const std::string &k) const { return kind == Object && obj.count(k) != 0; }
    const JsonValue &at(const std::string &k) const {
        if (kind != Object) throw std::runtime_error("JSON: not an object while looking up '" + k + "'");
        auto it = obj.find(k);
        if (it == obj.end()) throw std::runtime_error("JSON: missing key '" + k + "'");
        return it->second;
    }
    double as_number(const char *what) const {
        if (kind != Number) throw std::runtime_error(std::string("JSON: '") + what + "' is not a number");
        return num;
    }
    int as_int(const char *what) const {
        if (kind != Number || !is_int) throw std::runtime_error(std::string("JSON: '") + what + "' is not an integer");
        return (int)num;
    }
    const std::string &as_string(const char *what) const {
        if (kind != String) throw std::runtime_error(std::string("JSON: '") + what + "' is not a string");
        return str;
    }
    bool as_bool(const char *what) const {
        if (kind != Bool) throw std::runtime_error(std::string("JSON: '") + what + "' is not a bool");
        return b;
    }
};

class JsonReader {
  public:
    explicit JsonReader(const std::string &text) : s_(text) {}
    JsonValue parse() {
        JsonValue v = value();
        skip();
        if (p_ != s_.size()) fail("trailing characters");
        return v;
    }

  private:
    const std::string &s_;
    size_t p_ = 0;

    [[noreturn]] void fail(const char *msg) const {
        throw std::runtime_error("JSON parse error at byte " + std::to_string(p_) + ": " + msg);
    }
    void skip() {
        while (p_ < s_.size() && std::isspace((unsigned char)s_[p_])) ++p_;
    }
    bool eat(char c) {
        skip();
        if (p_ < s_.size() && s_[p_] == c) { ++p_; return true; }
        return false;
    }
    void expect(char c) {
        if (!eat(c)) { std::string m = std::string("expected '") + c + "'"; fail(m.c_str()); }
    }
    JsonValue value() {
        skip();
        if (p_ >= s_.size()) fail("unexpected end");
        char c = s_[p_];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') { JsonValue v; v.kind = JsonValue::String; v.str = string(); return v; }
        if (c == 't' || c == 'f' || c == 'n') return literal();
        return number();
    }
    JsonValue literal() {
        JsonValue v;
        if (s_.compare(p_, 4, "true") == 0) { v.kind = JsonValue::Bool; v.b = true; p_ += 4; }
        else if (s_.compare(p_, 5, "false") == 0) { v.kind = JsonValue::Bool; v.b = false; p_ += 5; }
        else if (s_.compare(p_, 4, "null") == 0) { v.kind = JsonValue::Null; p_ += 4; }
        else fail("bad literal");
        return v;
    }
    JsonValue number() {
        size_t start = p_;
        bool integral = true;
        if (p_ < s_.size() && (s_[p_] == '-' || s_[p_] == '+')) ++p_;
        while (p_ < s_.size()) {
            char c = s_[p_];
            if (std::isdigit((unsigned char)c)) { ++p_; }
            else if (c == '.' || c == 'e' || c == 'E' || c == '-' || c == '+') { integral = false; ++p_; }
            else break;
        }
        if (p_ == start) fail("bad number");
        // NaN / Infinity as written by Python's json.dump are not accepted (neither does rapidjson).
        JsonValue v;
        v.kind = JsonValue::Number;
        v.is_int = integral;
        v.num = std::strtod(s_.substr(start, p_ - start).c_str(), nullptr);
        return v;
    }
    std::string string() {
        expect('"');
        std::string out;
        while (p_ < s_.size() && s_[p_] != '"') {
            char c = s_[p_++];
            if (c == '\\') {
                if (p_ >= s_.size()) fail("bad escape");
                char e = s_[p_++];
                switch (e) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {  // only the ASCII range is needed for file paths
                        if (p_ + 4 > s_.size()) fail("bad \\u escape");
                        unsigned code = (unsigned)std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16);
                        p_ += 4;
                        out += (code < 0x80) ? (char)code : '?';
                        break;
                    }
                    default: out += e;  // \" \\ \/
                }
            } else {
                out += c;
            }
        }
        if (p_ >= s_.size()) fail("unterminated string");
        ++p_;
        return out;
    }
    JsonValue array() {
        JsonValue v;
        v.kind = JsonValue::Array;
        expect('[');
        if (eat(']')) return v;
        do { v.arr.push_back(value()); } while (eat(','));
        expect(']');
        return v;
    }
    JsonValue object() {
        JsonValue v;
        v.kind = JsonValue::Object;
        expect('{');
        if (eat('}')) return v;
        do {
            skip();
            std::string k = string();
            expect(':');
            v.obj[k] = value();
        } while (eat(','));
        expect('}');
        return v;
    }
};

}  // namespace sepfwi
