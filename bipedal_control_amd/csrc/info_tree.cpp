#include "info_tree.h"

#include <cstdlib>
#include <fstream>
#include <stdexcept>

namespace bpmpc {

namespace {
enum class Tok { Word, Open, Close };
struct Token { Tok kind; std::string text; };

// One physical line -> tokens.  ';' starts a comment; "..." may hold blanks; braces are their own tokens.
std::vector<Token> split_line(const std::string& line) {
  std::vector<Token> out;
  size_t i = 0;
  const size_t n = line.size();
  auto blank = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; };
  while (i < n) {
    const char c = line[i];
    if (blank(c)) { ++i; continue; }
    if (c == ';') break;
    if (c == '{') { out.push_back({Tok::Open, "{"}); ++i; continue; }
    if (c == '}') { out.push_back({Tok::Close, "}"}); ++i; continue; }
    std::string w;
    if (c == '"') {
      ++i;
      while (i < n && line[i] != '"') {
        if (line[i] == '\\' && i + 1 < n) ++i;
        w.push_back(line[i++]);
      }
      ++i;
    } else {
      while (i < n && !blank(line[i]) && line[i] != ';') w.push_back(line[i++]);
    }
    out.push_back({Tok::Word, w});
  }
  return out;
}
}  // namespace

std::unique_ptr<InfoNode> read_info_file(const std::string& path) {
  std::ifstream in(path);
  if (!in) throw std::runtime_error("cannot open INFO file: " + path);
  auto root = std::make_unique<InfoNode>();
  std::vector<InfoNode*> open{root.get()};
  InfoNode* recent = nullptr;  // node created by the latest key
  std::string line;
  while (std::getline(in, line)) {
    bool want_data = false;  // a key on this line still waits for its data token
    for (const Token& t : split_line(line)) {
      switch (t.kind) {
        case Tok::Open:
          if (!recent) throw std::runtime_error("INFO: '{' without a key in " + path);
          open.push_back(recent);
          recent = nullptr;
          want_data = false;
          break;
        case Tok::Close:
          if (open.size() == 1) throw std::runtime_error("INFO: unbalanced '}' in " + path);
          open.pop_back();
          recent = nullptr;
          want_data = false;
          break;
        case Tok::Word:
          if (want_data) {
            recent->data = t.text;
            want_data = false;
          } else {
            open.back()->children.emplace_back(t.text, std::make_unique<InfoNode>());
            recent = open.back()->children.back().second.get();
            want_data = true;
          }
          break;
      }
    }
  }
  if (open.size() != 1) throw std::runtime_error("INFO: unbalanced '{' in " + path);
  return root;
}

const InfoNode* InfoNode::find(const std::string& dotted_path) const {
  const InfoNode* node = this;
  size_t pos = 0;
  while (pos <= dotted_path.size()) {
    const size_t dot = dotted_path.find('.', pos);
    const std::string part = dotted_path.substr(pos, dot == std::string::npos ? std::string::npos : dot - pos);
    const InfoNode* next = nullptr;
    for (const auto& kv : node->children)
      if (kv.first == part) { next = kv.second.get(); break; }
    if (!next) return nullptr;
    node = next;
    if (dot == std::string::npos) break;
    pos = dot + 1;
  }
  return node;
}

bool InfoNode::get(const std::string& path, std::string* out) const {
  const InfoNode* n = find(path);
  if (!n) return false;
  *out = n->data;
  return true;
}
bool InfoNode::get(const std::string& path, double* out) const {
  const InfoNode* n = find(path);
  if (!n || n->data.empty()) return false;
  char* end = nullptr;
  const double v = std::strtod(n->data.c_str(), &end);
  if (end == n->data.c_str()) return false;
  *out = v;
  return true;
}
bool InfoNode::get(const std::string& path, int* out) const {
  double v;
  if (!get(path, &v)) return false;
  *out = static_cast<int>(v);
  return true;
}

std::vector<double> load_matrix(const InfoNode& root, const std::string& name, int rows, int cols) {
  double scaling = 1.0, fallback = 0.0;
  root.get(name + ".scaling", &scaling);
  root.get(name + ".default", &fallback);
  std::vector<double> m(static_cast<size_t>(rows) * cols);
  int missing = 0;
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < cols; ++j) {
      double a;
      if (!root.get(name + ".(" + std::to_string(i) + "," + std::to_string(j) + ")", &a)) { a = fallback; ++missing; }
      m[static_cast<size_t>(i) * cols + j] = scaling * a;
    }
  if (missing == rows * cols) throw std::runtime_error("INFO: could not load matrix \"" + name + "\"");
  return m;
}

std::vector<double> load_scalar_list(const InfoNode& root, const std::string& name) {
  std::vector<double> v;
  double a;
  while (root.get(name + ".[" + std::to_string(v.size()) + "]", &a)) v.push_back(a);
  return v;
}
std::vector<std::string> load_string_list(const InfoNode& root, const std::string& name) {
  std::vector<std::string> v;
  std::string s;
  while (root.get(name + ".[" + std::to_string(v.size()) + "]", &s)) v.push_back(s);
  return v;
}

}  // namespace bpmpc
