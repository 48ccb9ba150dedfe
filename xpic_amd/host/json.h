// json.h -- minimal JSON reader for the xpic configuration surface (objects keep key order, like the
// reference's nlohmann::ordered_json; src/utils/configuration.h:14).  No external dependency by design:
// nothing can be fetched on the build or GPU boxes.
#pragma once

#include <cctype>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace xjson {

struct Value {
  enum Type { Null, Bool, Number, String, Array, Object } type = Null;
  bool b = false;
  double num = 0;
  std::string str;
  std::vector<Value> arr;
  std::vector<std::pair<std::string, Value>> obj;

  bool is_string() const { return type == String; }
  bool is_number() const { return type == Number; }
  bool is_array() const { return type == Array; }
  bool is_object() const { return type == Object; }
  bool contains(const std::string& k) const { return find(k) != nullptr; }
  const Value* find(const std::string& k) const
  {
    for (auto& kv : obj)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
  const Value& at(const std::string& k) const
  {
    const Value* v = find(k);
    if (!v) throw std::runtime_error("json: key not found: " + k);
    return *v;
  }
  double as_double() const
  {
    if (type != Number) throw std::runtime_error("json: number expected");
    return num;
  }
  int as_int() const { return (int)as_double(); }
  const std::string& as_string() const
  {
    if (type != String) throw std::runtime_error("json: string expected");
    return str;
  }
  bool as_bool() const
  {
    if (type != Bool) throw std::runtime_error("json: bool expected");
    return b;
  }
};

class Parser {
public:
  explicit Parser(const std::string& s) : s_(s) {}
  Value parse()
  {
    Value v = value();
    ws();
    if (i_ != s_.size()) fail("trailing characters");
    return v;
  }

private:
  const std::string& s_;
  size_t i_ = 0;
  [[noreturn]] void fail(const std::string& m) { throw std::runtime_error("json parse error at " + std::to_string(i_) + ": " + m); }
  void ws()
  {
    while (i_ < s_.size() && std::isspace((unsigned char)s_[i_])) ++i_;
  }
  Value value()
  {
    ws();
    if (i_ >= s_.size()) fail("unexpected end");
    char c = s_[i_];
    if (c == '{') return object();
    if (c == '[') return array();
    if (c == '"') { Value v; v.type = Value::String; v.str = string(); return v; }
    if (s_.compare(i_, 4, "true") == 0) { i_ += 4; Value v; v.type = Value::Bool; v.b = true; return v; }
    if (s_.compare(i_, 5, "false") == 0) { i_ += 5; Value v; v.type = Value::Bool; v.b = false; return v; }
    if (s_.compare(i_, 4, "null") == 0) { i_ += 4; return Value(); }
    char* end = nullptr;
    double d = std::strtod(s_.c_str() + i_, &end);
    if (end == s_.c_str() + i_) fail("value expected");
    i_ = end - s_.c_str();
    Value v; v.type = Value::Number; v.num = d;
    return v;
  }
  std::string string()
  {
    std::string out;
    ++i_;
    while (i_ < s_.size() && s_[i_] != '"') {
      if (s_[i_] == '\\' && i_ + 1 < s_.size()) {
        char e = s_[++i_];
        out += e == 'n' ? '\n' : (e == 't' ? '\t' : e);
      }
      else out += s_[i_];
      ++i_;
    }
    if (i_ >= s_.size()) fail("unterminated string");
    ++i_;
    return out;
  }
  Value array()
  {
    Value v; v.type = Value::Array;
    ++i_;
    ws();
    if (s_[i_] == ']') { ++i_; return v; }
    for (;;) {
      v.arr.push_back(value());
      ws();
      if (s_[i_] == ',') { ++i_; continue; }
      if (s_[i_] == ']') { ++i_; return v; }
      fail("',' or ']' expected");
    }
  }
  Value object()
  {
    Value v; v.type = Value::Object;
    ++i_;
    ws();
    if (s_[i_] == '}') { ++i_; return v; }
    for (;;) {
      ws();
      if (s_[i_] != '"') fail("key expected");
      std::string k = string();
      ws();
      if (s_[i_] != ':') fail("':' expected");
      ++i_;
      v.obj.emplace_back(k, value());
      ws();
      if (s_[i_] == ',') { ++i_; continue; }
      if (s_[i_] == '}') { ++i_; return v; }
      fail("',' or '}' expected");
    }
  }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace xjson
