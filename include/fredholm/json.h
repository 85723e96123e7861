// json.h -- minimal JSON reader (RFC 8259) for the glTF loader of the drop-in Scene.  The reference parses glTF with tinygltf
// (externals/tinygltf, an empty submodule in the checkout); this is an independent recursive-descent parser: objects, arrays,
// strings with escapes (\uXXXX -> UTF-8, surrogate pairs included), numbers via strtod, true / false / null.
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace fredholm::json {

struct Value {
  enum Type { Null, Bool, Number, String, Array, Object } type = Null;
  bool b = false;
  double num = 0.0;
  std::string str;
  std::vector<Value> arr;
  std::map<std::string, Value> obj;

  bool has(const std::string& k) const { return type == Object && obj.count(k) != 0; }
  const Value& at(const std::string& k) const
  {
    if (type != Object) throw std::runtime_error("json: not an object (looking up '" + k + "')");
    const auto it = obj.find(k);
    if (it == obj.end()) throw std::runtime_error("json: missing key '" + k + "'");
    return it->second;
  }
  const Value& at(size_t i) const
  {
    if (type != Array || i >= arr.size()) throw std::runtime_error("json: array index out of range");
    return arr[i];
  }
  size_t size() const { return type == Array ? arr.size() : (type == Object ? obj.size() : 0); }
  double number() const { if (type != Number) throw std::runtime_error("json: not a number"); return num; }
  int integer() const { return int(number()); }
  const std::string& string() const { if (type != String) throw std::runtime_error("json: not a string"); return str; }
  double number_or(const std::string& k, double dflt) const { return has(k) ? at(k).number() : dflt; }
  int integer_or(const std::string& k, int dflt) const { return has(k) ? at(k).integer() : dflt; }
};

class Parser {
 public:
  explicit Parser(const std::string& text) : m_s(text) {}
  Value parse()
  {
    Value v = value();
    ws();
    if (m_p != m_s.size()) fail("trailing characters");
    return v;
  }

 private:
  const std::string& m_s;
  size_t m_p = 0;

  [[noreturn]] void fail(const char* what) const { throw std::runtime_error(std::string("json: ") + what + " at offset " + std::to_string(m_p)); }
  void ws() { while (m_p < m_s.size() && (m_s[m_p] == ' ' || m_s[m_p] == '\t' || m_s[m_p] == '\n' || m_s[m_p] == '\r')) ++m_p; }
  bool lit(const char* w)
  {
    const size_t n = std::char_traits<char>::length(w);
    if (m_s.compare(m_p, n, w) == 0) { m_p += n; return true; }
    return false;
  }
  static void utf8(std::string& out, unsigned cp)
  {
    if (cp < 0x80) out.push_back(char(cp));
    else if (cp < 0x800) { out.push_back(char(0xc0 | (cp >> 6))); out.push_back(char(0x80 | (cp & 0x3f))); }
    else if (cp < 0x10000) { out.push_back(char(0xe0 | (cp >> 12))); out.push_back(char(0x80 | ((cp >> 6) & 0x3f))); out.push_back(char(0x80 | (cp & 0x3f))); }
    else { out.push_back(char(0xf0 | (cp >> 18))); out.push_back(char(0x80 | ((cp >> 12) & 0x3f))); out.push_back(char(0x80 | ((cp >> 6) & 0x3f))); out.push_back(char(0x80 | (cp & 0x3f))); }
  }
  unsigned hex4()
  {
    if (m_p + 4 > m_s.size()) fail("truncated \\u escape");
    unsigned v = 0;
    for (int i = 0; i < 4; ++i) {
      const char c = m_s[m_p++];
      v <<= 4;
      if (c >= '0' && c <= '9') v |= unsigned(c - '0');
      else if (c >= 'a' && c <= 'f') v |= unsigned(c - 'a' + 10);
      else if (c >= 'A' && c <= 'F') v |= unsigned(c - 'A' + 10);
      else fail("bad \\u escape");
    }
    return v;
  }
  std::string string_body()
  {
    std::string out;
    ++m_p;  // opening quote
    for (;;) {
      if (m_p >= m_s.size()) fail("unterminated string");
      const char c = m_s[m_p++];
      if (c == '"') return out;
      if (c != '\\') { out.push_back(c); continue; }
      if (m_p >= m_s.size()) fail("unterminated escape");
      const char e = m_s[m_p++];
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
          unsigned cp = hex4();
          if (cp >= 0xd800 && cp < 0xdc00 && m_s.compare(m_p, 2, "\\u") == 0) {
            m_p += 2;
            const unsigned lo = hex4();
            cp = 0x10000 + ((cp - 0xd800) << 10) + (lo - 0xdc00);
          }
          utf8(out, cp);
          break;
        }
        default: fail("bad escape");
      }
    }
  }
  Value value()
  {
    ws();
    if (m_p >= m_s.size()) fail("unexpected end");
    Value v;
    const char c = m_s[m_p];
    if (c == '{') {
      v.type = Value::Object;
      ++m_p;
      ws();
      if (m_p < m_s.size() && m_s[m_p] == '}') { ++m_p; return v; }
      for (;;) {
        ws();
        if (m_p >= m_s.size() || m_s[m_p] != '"') fail("expected a key");
        std::string k = string_body();
        ws();
        if (m_p >= m_s.size() || m_s[m_p] != ':') fail("expected ':'");
        ++m_p;
        v.obj[k] = value();
        ws();
        if (m_p < m_s.size() && m_s[m_p] == ',') { ++m_p; continue; }
        if (m_p < m_s.size() && m_s[m_p] == '}') { ++m_p; return v; }
        fail("expected ',' or '}'");
      }
    }
    if (c == '[') {
      v.type = Value::Array;
      ++m_p;
      ws();
      if (m_p < m_s.size() && m_s[m_p] == ']') { ++m_p; return v; }
      for (;;) {
        v.arr.push_back(value());
        ws();
        if (m_p < m_s.size() && m_s[m_p] == ',') { ++m_p; continue; }
        if (m_p < m_s.size() && m_s[m_p] == ']') { ++m_p; return v; }
        fail("expected ',' or ']'");
      }
    }
    if (c == '"') { v.type = Value::String; v.str = string_body(); return v; }
    if (lit("true")) { v.type = Value::Bool; v.b = true; return v; }
    if (lit("false")) { v.type = Value::Bool; v.b = false; return v; }
    if (lit("null")) return v;
    if (c == '-' || (c >= '0' && c <= '9')) {
      const char* begin = m_s.c_str() + m_p;
      char* end = nullptr;
      v.type = Value::Number;
      v.num = std::strtod(begin, &end);
      if (end == begin) fail("bad number");
      m_p += size_t(end - begin);
      return v;
    }
    fail("unexpected character");
  }
};

inline Value parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace fredholm::json
