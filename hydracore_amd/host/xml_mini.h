// xml_mini.h -- tiny read-only XML DOM for HydraAPI scene-library state files (statex_*.xml).
//
// The reference walks the same files with pugixml (hydra_drv/RenderDriverRTE.cpp:160-396,
// PlainMaterialConverter.cpp:886-951); pugixml is not in this image, and only the subset the
// scene libraries use is needed: elements, quoted attributes, text, comments, the <?xml?> prolog,
// and several top-level elements in one file (textures_lib, materials_lib, ... are siblings).
#pragma once
#include <string>
#include <vector>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace hydra_host {

struct XmlNode {
  std::string name;
  std::string text;
  std::vector<std::pair<std::string, std::string>> attrs;
  std::vector<std::unique_ptr<XmlNode>> children;

  const XmlNode* child(const char* n) const {
    for (auto& c : children) if (c->name == n) return c.get();
    return nullptr;
  }
  std::vector<const XmlNode*> children_named(const char* n) const {
    std::vector<const XmlNode*> r;
    for (auto& c : children) if (c->name == n) r.push_back(c.get());
    return r;
  }
  bool has_attr(const char* n) const {
    for (auto& a : attrs) if (a.first == n) return true;
    return false;
  }
  const char* attr(const char* n, const char* dflt = "") const {
    for (auto& a : attrs) if (a.first == n) return a.second.c_str();
    return dflt;
  }
  int   attr_int(const char* n, int dflt = 0) const { return has_attr(n) ? atoi(attr(n)) : dflt; }
  float attr_float(const char* n, float dflt = 0.0f) const { return has_attr(n) ? strtof(attr(n), nullptr) : dflt; }
};

// null-safe navigation helpers (pugixml returns empty nodes; we return nullptr and tolerate it)
inline const XmlNode* xchild(const XmlNode* n, const char* name) { return n ? n->child(name) : nullptr; }
inline const char*    xattr(const XmlNode* n, const char* name, const char* d = "") { return n ? n->attr(name, d) : d; }
inline bool           xhas(const XmlNode* n, const char* name) { return n && n->has_attr(name); }
inline std::string    xtext(const XmlNode* n) { return n ? n->text : std::string(); }

class XmlParser {
public:
  explicit XmlParser(const std::string& src) : s(src), p(0) {}

  std::unique_ptr<XmlNode> parse_document() {
    auto root = std::make_unique<XmlNode>();
    root->name = "#document";
    while (true) {
      skip_misc();
      if (p >= s.size()) break;
      if (s[p] != '<') throw std::runtime_error("xml_mini: text outside of an element");
      root->children.push_back(parse_element());
    }
    return root;
  }

private:
  const std::string& s;
  size_t p;

  void skip_ws() { while (p < s.size() && (s[p] == ' ' || s[p] == '\t' || s[p] == '\r' || s[p] == '\n')) p++; }

  void skip_misc() {
    while (true) {
      skip_ws();
      if (s.compare(p, 4, "<!--") == 0) {
        size_t e = s.find("-->", p + 4);
        if (e == std::string::npos) throw std::runtime_error("xml_mini: unterminated comment");
        p = e + 3;
      } else if (s.compare(p, 2, "<?") == 0) {
        size_t e = s.find("?>", p + 2);
        if (e == std::string::npos) throw std::runtime_error("xml_mini: unterminated prolog");
        p = e + 2;
      } else
        break;
    }
  }

  std::string parse_name() {
    size_t b = p;
    while (p < s.size() && !strchr(" \t\r\n/>=", s[p])) p++;
    return s.substr(b, p - b);
  }

  static std::string unescape(const std::string& v) {
    if (v.find('&') == std::string::npos) return v;
    std::string r;
    for (size_t i = 0; i < v.size(); i++) {
      if (v[i] == '&') {
        if (!v.compare(i, 4, "&lt;")) { r += '<'; i += 3; }
        else if (!v.compare(i, 4, "&gt;")) { r += '>'; i += 3; }
        else if (!v.compare(i, 5, "&amp;")) { r += '&'; i += 4; }
        else if (!v.compare(i, 6, "&quot;")) { r += '"'; i += 5; }
        else if (!v.compare(i, 6, "&apos;")) { r += '\''; i += 5; }
        else r += v[i];
      } else
        r += v[i];
    }
    return r;
  }

  std::unique_ptr<XmlNode> parse_element() {
    auto n = std::make_unique<XmlNode>();
    p++;  // '<'
    n->name = parse_name();
    while (true) {
      skip_ws();
      if (p >= s.size()) throw std::runtime_error("xml_mini: unexpected end inside tag");
      if (s[p] == '/') {  // self closing
        p += 2;
        return n;
      }
      if (s[p] == '>') { p++; break; }
      std::string an = parse_name();
      skip_ws();
      if (s[p] != '=') throw std::runtime_error("xml_mini: attribute without value in <" + n->name + ">");
      p++;
      skip_ws();
      char q = s[p++];
      if (q != '"' && q != '\'') throw std::runtime_error("xml_mini: unquoted attribute");
      size_t e = s.find(q, p);
      if (e == std::string::npos) throw std::runtime_error("xml_mini: unterminated attribute");
      n->attrs.emplace_back(an, unescape(s.substr(p, e - p)));
      p = e + 1;
    }
    // content
    while (true) {
      size_t lt = s.find('<', p);
      if (lt == std::string::npos) throw std::runtime_error("xml_mini: unterminated element " + n->name);
      n->text += s.substr(p, lt - p);
      p = lt;
      if (s.compare(p, 4, "<!--") == 0) {
        size_t e = s.find("-->", p + 4);
        if (e == std::string::npos) throw std::runtime_error("xml_mini: unterminated comment");
        p = e + 3;
      } else if (s.compare(p, 2, "</") == 0) {
        size_t e = s.find('>', p);
        p = e + 1;
        break;
      } else
        n->children.push_back(parse_element());
    }
    // trim text
    size_t b = n->text.find_first_not_of(" \t\r\n");
    size_t e = n->text.find_last_not_of(" \t\r\n");
    n->text = (b == std::string::npos) ? std::string() : unescape(n->text.substr(b, e - b + 1));
    return n;
  }
};

}  // namespace hydra_host
