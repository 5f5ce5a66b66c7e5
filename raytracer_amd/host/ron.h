// Parser for the subset of RON (Rusty Object Notation, ron 0.8.0) that the
// reference's scene files use (scenes/*.ron, read at src/scene/loader.rs:50-52):
// optionally named structs `Name(field: v, ..)`, tuples / tuple variants
// `Name(v, ..)`, unit variants `Name`, lists `[..]`, maps `{ "k": v }`,
// strings, numbers with `_` digit separators, `//` and `/* */` comments and
// trailing commas.  The result is an untyped tree; typing happens in loader.cpp.
#pragma once
#include <string>
#include <utility>
#include <vector>

namespace rthost {

struct RonNode {
  enum Kind { NUMBER, STRING, TUPLE, STRUCT, LIST, MAP };
  Kind kind = NUMBER;
  double number = 0.0;
  bool is_integer = false;         // NUMBER written without '.', 'e' or 'E'
  std::string text;                // STRING value, or variant / struct name (may be empty)
  std::vector<RonNode> items;      // TUPLE, LIST
  std::vector<std::pair<std::string, RonNode>> fields;  // STRUCT (by name), MAP (string keys), file order
  int line = 0;

  bool is_unit() const { return kind == TUPLE && items.empty() && !text.empty(); }
  const RonNode* field(const char* name) const;
};

// Throws std::runtime_error("line N: ...") on malformed input.
RonNode ron_parse(const std::string& text);

}  // namespace rthost
