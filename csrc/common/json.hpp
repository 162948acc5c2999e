// Minimal JSON value (parse + serialise) for the native control-plane binaries
// (picker service, operator).  Objects keep insertion order.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace vjson {

struct Value;
using Array = std::vector<Value>;
using Member = std::pair<std::string, Value>;
using Object = std::vector<Member>;

struct Value {
  enum Type { Null, Bool, Number, String, Arr, Obj } type = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::shared_ptr<Array> arr;
  std::shared_ptr<Object> obj;

  Value() = default;
  Value(std::nullptr_t) {}
  Value(bool v) : type(Bool), b(v) {}
  Value(int v) : type(Number), num(v) {}
  Value(int64_t v) : type(Number), num((double)v) {}
  Value(double v) : type(Number), num(v) {}
  Value(const char* s) : type(String), str(s) {}
  Value(std::string s) : type(String), str(std::move(s)) {}
  static Value array() {
    Value v;
    v.type = Arr;
    v.arr = std::make_shared<Array>();
    return v;
  }
  static Value object() {
    Value v;
    v.type = Obj;
    v.obj = std::make_shared<Object>();
    return v;
  }

  bool is_null() const { return type == Null; }
  bool is_obj() const { return type == Obj; }
  bool is_arr() const { return type == Arr; }
  bool is_str() const { return type == String; }
  bool is_num() const { return type == Number; }

  const Value* find(const std::string& k) const {
    if (type != Obj) return nullptr;
    for (auto& m : *obj)
      if (m.first == k) return &m.second;
    return nullptr;
  }
  // path lookup "a.b.c"; returns Null value when missing
  const Value& at(const std::string& path) const {
    static const Value nullv;
    const Value* cur = this;
    size_t pos = 0;
    while (pos <= path.size()) {
      size_t dot = path.find('.', pos);
      std::string k = path.substr(pos, dot == std::string::npos ? std::string::npos : dot - pos);
      cur = cur->find(k);
      if (!cur) return nullv;
      if (dot == std::string::npos) break;
      pos = dot + 1;
    }
    return *cur;
  }
  std::string s(const std::string& path, const std::string& def = "") const {
    const Value& v = at(path);
    return v.type == String ? v.str : def;
  }
  double n(const std::string& path, double def = 0) const {
    const Value& v = at(path);
    return v.type == Number ? v.num : def;
  }
  // integer view of a number, clamped to [lo, hi]; `def` when missing, not a number or not finite
  // (a double outside the target range must never reach an integer cast: that is undefined)
  int64_t i(const std::string& path, int64_t def = 0, int64_t lo = -(int64_t(1) << 53),
            int64_t hi = int64_t(1) << 53) const {
    const Value& v = at(path);
    if (v.type != Number || !std::isfinite(v.num)) return def;
    if (v.num <= (double)lo) return lo;
    if (v.num >= (double)hi) return hi;
    return (int64_t)v.num;
  }
  bool flag(const std::string& path, bool def = false) const {
    const Value& v = at(path);
    return v.type == Bool ? v.b : def;
  }
  Value& set(const std::string& k, Value v) {
    if (type != Obj) *this = object();
    for (auto& m : *obj)
      if (m.first == k) {
        m.second = std::move(v);
        return m.second;
      }
    obj->emplace_back(k, std::move(v));
    return obj->back().second;
  }
  Value& operator[](const std::string& k) {
    if (type != Obj) *this = object();
    for (auto& m : *obj)
      if (m.first == k) return m.second;
    obj->emplace_back(k, Value());
    return obj->back().second;
  }
  void push(Value v) {
    if (type != Arr) *this = array();
    arr->push_back(std::move(v));
  }
  size_t size() const { return type == Arr ? arr->size() : type == Obj ? obj->size() : 0; }
};

inline void escape(const std::string& s, std::string& out) {
  out += '"';
  for (unsigned char c : s) {
    switch (c) {
      case '"': out += "\\\""; break;
      case '\\': out += "\\\\"; break;
      case '\n': out += "\\n"; break;
      case '\r': out += "\\r"; break;
      case '\t': out += "\\t"; break;
      default:
        if (c < 0x20) {
          char buf[8];
          snprintf(buf, sizeof buf, "\\u%04x", c);
          out += buf;
        } else {
          out += (char)c;
        }
    }
  }
  out += '"';
}

inline void dump(const Value& v, std::string& out) {
  switch (v.type) {
    case Value::Null: out += "null"; break;
    case Value::Bool: out += v.b ? "true" : "false"; break;
    case Value::Number: {
      char buf[32];
      if (!std::isfinite(v.num)) {  // JSON has no NaN / Infinity
        out += "null";
        break;
      }
      if (v.num > -1e15 && v.num < 1e15 && v.num == std::floor(v.num))
        snprintf(buf, sizeof buf, "%lld", (long long)v.num);
      else
        snprintf(buf, sizeof buf, "%.17g", v.num);
      out += buf;
      break;
    }
    case Value::String: escape(v.str, out); break;
    case Value::Arr: {
      out += '[';
      bool first = true;
      for (auto& e : *v.arr) {
        if (!first) out += ',';
        first = false;
        dump(e, out);
      }
      out += ']';
      break;
    }
    case Value::Obj: {
      out += '{';
      bool first = true;
      for (auto& m : *v.obj) {
        if (!first) out += ',';
        first = false;
        escape(m.first, out);
        out += ':';
        dump(m.second, out);
      }
      out += '}';
      break;
    }
  }
}

inline std::string dump(const Value& v) {
  std::string s;
  dump(v, s);
  return s;
}

class Parser {
 public:
  explicit Parser(const std::string& s) : p_(s.data()), end_(s.data() + s.size()) {}
  Value parse() {
    Value v = value();
    ws();
    if (p_ != end_) throw std::runtime_error("trailing characters after JSON value");
    return v;
  }

 private:
  // nesting deeper than this is refused: the parser recurses once per level, and a body of
  // nothing but '[' would otherwise turn a 64 MiB request into a stack overflow
  static constexpr int kMaxDepth = 192;
  const char* p_;
  const char* end_;
  int depth_ = 0;
  struct Nest {
    int& d;
    explicit Nest(int& depth) : d(depth) { ++d; }
    ~Nest() { --d; }
  };
  void ws() {
    while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
  }
  [[noreturn]] void fail(const char* m) { throw std::runtime_error(std::string("JSON: ") + m); }
  Value value() {
    ws();
    if (p_ >= end_) fail("unexpected end");
    char c = *p_;
    if (c == '{') return object();
    if (c == '[') return array();
    if (c == '"') return Value(string());
    if (c == 't' && end_ - p_ >= 4 && !memcmp(p_, "true", 4)) { p_ += 4; return Value(true); }
    if (c == 'f' && end_ - p_ >= 5 && !memcmp(p_, "false", 5)) { p_ += 5; return Value(false); }
    if (c == 'n' && end_ - p_ >= 4 && !memcmp(p_, "null", 4)) { p_ += 4; return Value(); }
    return number();
  }
  Value number() {
    // JSON grammar only: strtod alone would also take "inf", "nan", hex floats and a leading '+'
    const char* q = p_;
    if (q < end_ && *q == '-') ++q;
    if (q >= end_ || *q < '0' || *q > '9') fail("bad number");
    while (q < end_ && *q >= '0' && *q <= '9') ++q;
    if (q < end_ && *q == '.') {
      ++q;
      if (q >= end_ || *q < '0' || *q > '9') fail("bad number");
      while (q < end_ && *q >= '0' && *q <= '9') ++q;
    }
    if (q < end_ && (*q == 'e' || *q == 'E')) {
      ++q;
      if (q < end_ && (*q == '+' || *q == '-')) ++q;
      if (q >= end_ || *q < '0' || *q > '9') fail("bad number");
      while (q < end_ && *q >= '0' && *q <= '9') ++q;
    }
    if (q - p_ > 400) fail("number too long");
    std::string tok(p_, q);
    double d = strtod(tok.c_str(), nullptr);
    if (!std::isfinite(d)) d = d > 0 ? 1.7976931348623157e308 : -1.7976931348623157e308;
    p_ = q;
    return Value(d);
  }
  static void utf8(unsigned cp, std::string& out) {
    if (cp < 0x80) out += (char)cp;
    else if (cp < 0x800) { out += (char)(0xC0 | cp >> 6); out += (char)(0x80 | (cp & 0x3F)); }
    else if (cp < 0x10000) { out += (char)(0xE0 | cp >> 12); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
    else { out += (char)(0xF0 | cp >> 18); out += (char)(0x80 | ((cp >> 12) & 0x3F)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
  }
  unsigned hex4() {
    if (end_ - p_ < 4) fail("bad \\u escape");
    unsigned v = 0;
    for (int i = 0; i < 4; ++i) {
      char c = *p_++;
      v <<= 4;
      if (c >= '0' && c <= '9') v |= c - '0';
      else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
      else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
      else fail("bad hex digit");
    }
    return v;
  }
  std::string string() {
    ++p_;
    std::string out;
    while (p_ < end_ && *p_ != '"') {
      if (*p_ == '\\') {
        ++p_;
        if (p_ >= end_) fail("bad escape");
        char c = *p_++;
        switch (c) {
          case 'n': out += '\n'; break;
          case 't': out += '\t'; break;
          case 'r': out += '\r'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'u': {
            unsigned cp = hex4();
            if (cp >= 0xD800 && cp < 0xDC00) {
              // a high surrogate counts only with its low half; alone it becomes U+FFFD
              unsigned lo = 0;
              if (end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                const char* save = p_;
                p_ += 2;
                lo = hex4();
                if (lo < 0xDC00 || lo > 0xDFFF) {
                  p_ = save;
                  lo = 0;
                }
              }
              cp = lo ? 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00) : 0xFFFD;
            } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
              cp = 0xFFFD;
            }
            utf8(cp, out);
            break;
          }
          default: out += c;
        }
      } else {
        out += *p_++;
      }
    }
    if (p_ >= end_) fail("unterminated string");
    ++p_;
    return out;
  }
  Value array() {
    Nest nest(depth_);
    if (depth_ > kMaxDepth) fail("nested too deeply");
    ++p_;
    Value v = Value::array();
    ws();
    if (p_ < end_ && *p_ == ']') { ++p_; return v; }
    while (true) {
      v.arr->push_back(value());
      ws();
      if (p_ >= end_) fail("unterminated array");
      if (*p_ == ',') { ++p_; continue; }
      if (*p_ == ']') { ++p_; return v; }
      fail("expected , or ]");
    }
  }
  Value object() {
    Nest nest(depth_);
    if (depth_ > kMaxDepth) fail("nested too deeply");
    ++p_;
    Value v = Value::object();
    ws();
    if (p_ < end_ && *p_ == '}') { ++p_; return v; }
    while (true) {
      ws();
      if (p_ >= end_ || *p_ != '"') fail("expected object key");
      std::string k = string();
      ws();
      if (p_ >= end_ || *p_ != ':') fail("expected :");
      ++p_;
      v.obj->emplace_back(std::move(k), value());
      ws();
      if (p_ >= end_) fail("unterminated object");
      if (*p_ == ',') { ++p_; continue; }
      if (*p_ == '}') { ++p_; return v; }
      fail("expected , or }");
    }
  }
};

inline Value parse(const std::string& s) { return Parser(s).parse(); }

}  // namespace vjson
