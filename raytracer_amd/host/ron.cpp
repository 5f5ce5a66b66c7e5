#include "ron.h"

#include <cctype>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace rthost {

const RonNode* RonNode::field(const char* name) const {
  for (const auto& f : fields)
    if (f.first == name) return &f.second;
  return nullptr;
}

namespace {

struct Parser {
  const std::string& s;
  size_t pos = 0;
  int line = 1;

  explicit Parser(const std::string& text) : s(text) {}

  [[noreturn]] void fail(const std::string& what) const {
    throw std::runtime_error("line " + std::to_string(line) + ": " + what);
  }

  bool eof() const { return pos >= s.size(); }
  char peek() const { return eof() ? '\0' : s[pos]; }
  char get() {
    char c = s[pos++];
    if (c == '\n') ++line;
    return c;
  }

  void skip_ws() {
    for (;;) {
      while (!eof() && std::isspace((unsigned char)peek())) get();
      if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '/') {
        while (!eof() && peek() != '\n') get();
        continue;
      }
      if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '*') {
        int depth = 0;
        do {
          if (pos + 1 < s.size() && s[pos] == '/' && s[pos + 1] == '*') {
            ++depth;
            get();
            get();
          } else if (pos + 1 < s.size() && s[pos] == '*' && s[pos + 1] == '/') {
            --depth;
            get();
            get();
          } else if (eof()) {
            fail("unterminated block comment");
          } else {
            get();
          }
        } while (depth > 0);
        continue;
      }
      break;
    }
  }

  static bool ident_start(char c) { return std::isalpha((unsigned char)c) || c == '_'; }
  static bool ident_char(char c) { return std::isalnum((unsigned char)c) || c == '_'; }

  std::string parse_ident() {
    size_t b = pos;
    while (!eof() && ident_char(peek())) get();
    return s.substr(b, pos - b);
  }

  // True when the text at `pos` (after whitespace) is `identifier :` — a struct field.
  bool looks_like_field() {
    size_t save = pos;
    int save_line = line;
    bool yes = false;
    if (ident_start(peek())) {
      parse_ident();
      skip_ws();
      yes = peek() == ':';
    }
    pos = save;
    line = save_line;
    return yes;
  }

  RonNode parse_string() {
    RonNode n;
    n.kind = RonNode::STRING;
    n.line = line;
    get();  // opening quote
    for (;;) {
      if (eof()) fail("unterminated string");
      char c = get();
      if (c == '"') break;
      if (c == '\\') {
        if (eof()) fail("unterminated escape");
        char e = get();
        switch (e) {
          case 'n': n.text.push_back('\n'); break;
          case 't': n.text.push_back('\t'); break;
          case 'r': n.text.push_back('\r'); break;
          case '0': n.text.push_back('\0'); break;
          case '\\': n.text.push_back('\\'); break;
          case '"': n.text.push_back('"'); break;
          case '\'': n.text.push_back('\''); break;
          default: fail(std::string("unsupported escape \\") + e);
        }
      } else {
        n.text.push_back(c);
      }
    }
    return n;
  }

  RonNode parse_number() {
    RonNode n;
    n.kind = RonNode::NUMBER;
    n.line = line;
    std::string digits;
    bool integer = true;
    if (peek() == '+' || peek() == '-') digits.push_back(get());
    bool any = false;
    while (!eof()) {
      char c = peek();
      if (std::isdigit((unsigned char)c)) {
        digits.push_back(get());
        any = true;
      } else if (c == '_' && any) {
        get();  // digit separator
      } else if (c == '.') {
        integer = false;
        digits.push_back(get());
      } else if (c == 'e' || c == 'E') {
        integer = false;
        digits.push_back(get());
        if (peek() == '+' || peek() == '-') digits.push_back(get());
      } else {
        break;
      }
    }
    if (!any) fail("expected a number");
    char* end = nullptr;
    n.number = std::strtod(digits.c_str(), &end);
    if (end == digits.c_str() || *end != '\0') fail("malformed number '" + digits + "'");
    n.is_integer = integer;
    return n;
  }

  // Contents of (...) after the opening parenthesis was consumed.
  void parse_paren_body(RonNode& n) {
    skip_ws();
    if (peek() == ')') {
      get();
      n.kind = RonNode::TUPLE;
      return;
    }
    if (looks_like_field()) {
      n.kind = RonNode::STRUCT;
      for (;;) {
        skip_ws();
        if (peek() == ')') {
          get();
          return;
        }
        if (!ident_start(peek())) fail("expected a field name");
        std::string key = parse_ident();
        skip_ws();
        if (peek() != ':') fail("expected ':' after field '" + key + "'");
        get();
        RonNode v = parse_value();
        n.fields.emplace_back(std::move(key), std::move(v));
        skip_ws();
        if (peek() == ',') {
          get();
        } else if (peek() == ')') {
          get();
          return;
        } else {
          fail("expected ',' or ')' in struct");
        }
      }
    }
    n.kind = RonNode::TUPLE;
    for (;;) {
      skip_ws();
      if (peek() == ')') {
        get();
        return;
      }
      n.items.push_back(parse_value());
      skip_ws();
      if (peek() == ',') {
        get();
      } else if (peek() == ')') {
        get();
        return;
      } else {
        fail("expected ',' or ')' in tuple");
      }
    }
  }

  RonNode parse_value() {
    skip_ws();
    if (eof()) fail("unexpected end of input");
    char c = peek();
    if (c == '"') return parse_string();
    if (std::isdigit((unsigned char)c) || c == '-' || c == '+' || c == '.') return parse_number();
    if (c == '(') {
      RonNode n;
      n.line = line;
      get();
      parse_paren_body(n);
      return n;
    }
    if (c == '[') {
      RonNode n;
      n.kind = RonNode::LIST;
      n.line = line;
      get();
      for (;;) {
        skip_ws();
        if (peek() == ']') {
          get();
          return n;
        }
        n.items.push_back(parse_value());
        skip_ws();
        if (peek() == ',') {
          get();
        } else if (peek() == ']') {
          get();
          return n;
        } else {
          fail("expected ',' or ']' in list");
        }
      }
    }
    if (c == '{') {
      RonNode n;
      n.kind = RonNode::MAP;
      n.line = line;
      get();
      for (;;) {
        skip_ws();
        if (peek() == '}') {
          get();
          return n;
        }
        if (peek() != '"') fail("expected a string key in map");
        RonNode k = parse_string();
        skip_ws();
        if (peek() != ':') fail("expected ':' after map key");
        get();
        RonNode v = parse_value();
        n.fields.emplace_back(std::move(k.text), std::move(v));
        skip_ws();
        if (peek() == ',') {
          get();
        } else if (peek() == '}') {
          get();
          return n;
        } else {
          fail("expected ',' or '}' in map");
        }
      }
    }
    if (ident_start(c)) {
      RonNode n;
      n.line = line;
      n.text = parse_ident();
      // a variant payload may be separated from its name by spaces: `Sphere (`
      size_t save = pos;
      int save_line = line;
      skip_ws();
      if (peek() == '(') {
        get();
        parse_paren_body(n);
      } else {
        pos = save;
        line = save_line;
        n.kind = RonNode::TUPLE;  // unit variant
      }
      return n;
    }
    fail(std::string("unexpected character '") + c + "'");
  }
};

}  // namespace

RonNode ron_parse(const std::string& text) {
  Parser p(text);
  RonNode root = p.parse_value();
  p.skip_ws();
  if (!p.eof()) p.fail("trailing characters after the scene");
  return root;
}

}  // namespace rthost
