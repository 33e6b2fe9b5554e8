#include "urdf_tree.h"

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace bpmpc {

namespace {

struct XmlElement {
  std::string tag;
  std::map<std::string, std::string> attr;
  std::vector<XmlElement> kids;
  const XmlElement* child(const std::string& t) const {
    for (const auto& k : kids)
      if (k.tag == t) return &k;
    return nullptr;
  }
  std::string get(const std::string& key, const std::string& fallback) const {
    auto it = attr.find(key);
    return it == attr.end() ? fallback : it->second;
  }
};

// Recursive-descent reader for the XML subset URDF files use: prolog, comments, elements, attributes.
class XmlReader {
 public:
  explicit XmlReader(std::string text) : s_(std::move(text)) {}
  XmlElement document() {
    skip_misc();
    XmlElement root = element();
    return root;
  }

 private:
  std::string s_;
  size_t p_ = 0;

  [[noreturn]] void fail(const std::string& why) const { throw std::runtime_error("URDF/XML parse error: " + why); }
  bool starts(const char* lit) const { return s_.compare(p_, std::char_traits<char>::length(lit), lit) == 0; }
  void skip_ws() { while (p_ < s_.size() && std::isspace(static_cast<unsigned char>(s_[p_]))) ++p_; }
  void skip_misc() {  // whitespace, <?...?>, <!-- ... -->, <!DOCTYPE ...>
    for (;;) {
      skip_ws();
      if (starts("<?")) { const size_t e = s_.find("?>", p_); if (e == std::string::npos) fail("unterminated <?"); p_ = e + 2; }
      else if (starts("<!--")) { const size_t e = s_.find("-->", p_); if (e == std::string::npos) fail("unterminated comment"); p_ = e + 3; }
      else if (starts("<!")) { const size_t e = s_.find('>', p_); if (e == std::string::npos) fail("unterminated <!"); p_ = e + 1; }
      else return;
    }
  }
  std::string name() {
    const size_t b = p_;
    while (p_ < s_.size() && (std::isalnum(static_cast<unsigned char>(s_[p_])) || s_[p_] == '_' || s_[p_] == ':' || s_[p_] == '-' || s_[p_] == '.')) ++p_;
    if (p_ == b) fail("name expected");
    return s_.substr(b, p_ - b);
  }
  XmlElement element() {
    if (p_ >= s_.size() || s_[p_] != '<') fail("'<' expected");
    ++p_;
    XmlElement e;
    e.tag = name();
    for (;;) {
      skip_ws();
      if (p_ >= s_.size()) fail("unexpected end inside <" + e.tag);
      if (starts("/>")) { p_ += 2; return e; }
      if (s_[p_] == '>') { ++p_; break; }
      const std::string key = name();
      skip_ws();
      if (s_[p_] != '=') fail("'=' expected after attribute " + key);
      ++p_;
      skip_ws();
      const char q = s_[p_];
      if (q != '"' && q != '\'') fail("quoted attribute value expected");
      const size_t e2 = s_.find(q, p_ + 1);
      if (e2 == std::string::npos) fail("unterminated attribute value");
      e.attr[key] = s_.substr(p_ + 1, e2 - p_ - 1);
      p_ = e2 + 1;
    }
    for (;;) {  // content
      const size_t lt = s_.find('<', p_);
      if (lt == std::string::npos) fail("unterminated <" + e.tag + ">");
      p_ = lt;
      if (starts("<!--")) { const size_t c = s_.find("-->", p_); if (c == std::string::npos) fail("unterminated comment"); p_ = c + 3; continue; }
      if (starts("<![CDATA[")) { const size_t c = s_.find("]]>", p_); if (c == std::string::npos) fail("unterminated CDATA"); p_ = c + 3; continue; }
      if (starts("</")) {
        p_ += 2;
        const std::string close = name();
        if (close != e.tag) fail("mismatched </" + close + "> for <" + e.tag + ">");
        skip_ws();
        if (s_[p_] != '>') fail("'>' expected");
        ++p_;
        return e;
      }
      e.kids.push_back(element());
    }
  }
};

void parse_triplet(const std::string& text, double out[3]) {
  std::istringstream is(text);
  for (int i = 0; i < 3; ++i) {
    std::string w;
    if (!(is >> w)) throw std::runtime_error("URDF: expected three numbers in \"" + text + "\"");
    out[i] = std::strtod(w.c_str(), nullptr);
  }
}
double parse_number(const std::string& text) { return std::strtod(text.c_str(), nullptr); }

}  // namespace

void rpy_to_matrix(const double rpy[3], double R[9]) {
  const double cr = std::cos(rpy[0]), sr = std::sin(rpy[0]), cp = std::cos(rpy[1]), sp = std::sin(rpy[1]), cy = std::cos(rpy[2]), sy = std::sin(rpy[2]);
  // explicit product Rz*Ry*Rx with the same association order as (Rz*Ry)*Rx
  const double zy[9] = {cy * cp, -sy, cy * sp, sy * cp, cy, sy * sp, -sp, 0.0, cp};
  const double rx[9] = {1, 0, 0, 0, cr, -sr, 0, sr, cr};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = zy[3 * i] * rx[j] + zy[3 * i + 1] * rx[3 + j] + zy[3 * i + 2] * rx[6 + j];
}

UrdfRobot read_urdf_file(const std::string& path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("cannot open URDF file: " + path);
  std::stringstream buf;
  buf << in.rdbuf();
  XmlReader reader(buf.str());
  const XmlElement root = reader.document();
  if (root.tag != "robot") throw std::runtime_error("URDF: root element is <" + root.tag + ">, expected <robot>");
  UrdfRobot robot;
  for (const XmlElement& e : root.kids) {
    if (e.tag == "link") {
      UrdfLink l;
      l.name = e.get("name", "");
      if (const XmlElement* ine = e.child("inertial")) {
        const XmlElement* mass = ine->child("mass");
        const XmlElement* it = ine->child("inertia");
        if (!mass || !it) throw std::runtime_error("URDF: <inertial> of link " + l.name + " lacks mass/inertia");
        l.mass = parse_number(mass->get("value", "0"));
        double rpy[3] = {0, 0, 0};
        if (const XmlElement* org = ine->child("origin")) {
          parse_triplet(org->get("xyz", "0 0 0"), l.com);
          parse_triplet(org->get("rpy", "0 0 0"), rpy);
        }
        const double ixx = parse_number(it->get("ixx", "0")), ixy = parse_number(it->get("ixy", "0")), ixz = parse_number(it->get("ixz", "0")),
                     iyy = parse_number(it->get("iyy", "0")), iyz = parse_number(it->get("iyz", "0")), izz = parse_number(it->get("izz", "0"));
        const double I0[9] = {ixx, ixy, ixz, ixy, iyy, iyz, ixz, iyz, izz};
        double R[9], T[9];
        rpy_to_matrix(rpy, R);
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) T[3 * i + j] = R[3 * i] * I0[j] + R[3 * i + 1] * I0[3 + j] + R[3 * i + 2] * I0[6 + j];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) l.inertia[3 * i + j] = T[3 * i] * R[3 * j] + T[3 * i + 1] * R[3 * j + 1] + T[3 * i + 2] * R[3 * j + 2];
      }
      robot.links.push_back(l);
    } else if (e.tag == "joint") {
      UrdfJoint j;
      j.name = e.get("name", "");
      j.type = e.get("type", "fixed");
      const XmlElement* par = e.child("parent");
      const XmlElement* chi = e.child("child");
      if (!par || !chi) throw std::runtime_error("URDF: joint " + j.name + " lacks parent/child");
      j.parent = par->get("link", "");
      j.child = chi->get("link", "");
      if (const XmlElement* org = e.child("origin")) {
        parse_triplet(org->get("xyz", "0 0 0"), j.xyz);
        parse_triplet(org->get("rpy", "0 0 0"), j.rpy);
      }
      if (const XmlElement* ax = e.child("axis")) parse_triplet(ax->get("xyz", "1 0 0"), j.axis);
      robot.joints.push_back(j);
    }
  }
  if (robot.links.empty()) throw std::runtime_error("URDF: no links in " + path);
  return robot;
}

}  // namespace bpmpc
