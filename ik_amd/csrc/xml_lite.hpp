// xml_lite.hpp -- the small XML subset a URDF needs: elements, attributes, comments,
// processing instructions, DOCTYPE, CDATA and text (both skipped).  No dependencies.
// Stands in for the urdfdom/tinyxml parse behind pinocchio::urdf::buildModelFromXML
// (called at reference ik_ros/src/cassie.cpp:34-35); neither library exists in this image.
#pragma once
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace ikgpu {
namespace xml {

struct Element {
    std::string tag;
    std::vector<std::pair<std::string, std::string>> attrs;
    std::vector<std::unique_ptr<Element>> children;

    const std::string *attr(const char *name) const {
        for (const auto &a : attrs)
            if (a.first == name) return &a.second;
        return nullptr;
    }
    const Element *child(const char *name) const {
        for (const auto &c : children)
            if (c->tag == name) return c.get();
        return nullptr;
    }
};

class Parser {
   public:
    Parser(const char *s, size_t n) : s_(s), n_(n) {}

    std::unique_ptr<Element> parse_document() {
        skip_misc();
        if (eof() || s_[i_] != '<') fail("expected root element");
        auto root = parse_element();
        skip_misc();
        return root;
    }

   private:
    const char *s_;
    size_t n_, i_ = 0;

    bool eof() const { return i_ >= n_; }
    [[noreturn]] void fail(const std::string &what) const {
        size_t line = 1;
        for (size_t k = 0; k < i_ && k < n_; ++k) line += (s_[k] == '\n');
        throw std::runtime_error("XML parse error at line " + std::to_string(line) + ": " + what);
    }
    bool starts(const char *lit) const {
        size_t l = std::strlen(lit);
        return i_ + l <= n_ && std::memcmp(s_ + i_, lit, l) == 0;
    }
    void skip_until(const char *lit) {
        size_t l = std::strlen(lit);
        while (i_ + l <= n_) {
            if (std::memcmp(s_ + i_, lit, l) == 0) { i_ += l; return; }
            ++i_;
        }
        fail(std::string("unterminated construct, expected ") + lit);
    }
    static bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; }
    void skip_space() { while (!eof() && is_space(s_[i_])) ++i_; }
    // whitespace, comments, <?...?>, <!DOCTYPE ...>
    void skip_misc() {
        for (;;) {
            skip_space();
            if (starts("<!--")) { i_ += 4; skip_until("-->"); }
            else if (starts("<?")) { i_ += 2; skip_until("?>"); }
            else if (starts("<!DOCTYPE")) { skip_until(">"); }
            else return;
        }
    }
    static bool is_name_char(char c) {
        return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-' ||
               c == ':' || c == '.';
    }
    std::string parse_name() {
        size_t b = i_;
        while (!eof() && is_name_char(s_[i_])) ++i_;
        if (b == i_) fail("expected a name");
        return std::string(s_ + b, i_ - b);
    }
    static std::string decode(const std::string &v) {
        if (v.find('&') == std::string::npos) return v;
        std::string o;
        for (size_t k = 0; k < v.size(); ++k) {
            if (v[k] != '&') { o += v[k]; continue; }
            static const struct { const char *ent; char ch; } tab[] = {
                {"&amp;", '&'}, {"&lt;", '<'}, {"&gt;", '>'}, {"&quot;", '"'}, {"&apos;", '\''}};
            bool hit = false;
            for (const auto &t : tab) {
                size_t l = std::strlen(t.ent);
                if (v.compare(k, l, t.ent) == 0) { o += t.ch; k += l - 1; hit = true; break; }
            }
            if (!hit) o += '&';
        }
        return o;
    }
    std::unique_ptr<Element> parse_element() {
        // at '<'
        ++i_;
        auto el = std::make_unique<Element>();
        el->tag = parse_name();
        for (;;) {
            skip_space();
            if (eof()) fail("unterminated start tag <" + el->tag);
            if (s_[i_] == '/') {
                if (!starts("/>")) fail("malformed empty-element tag");
                i_ += 2;
                return el;
            }
            if (s_[i_] == '>') { ++i_; break; }
            std::string key = parse_name();
            skip_space();
            if (eof() || s_[i_] != '=') fail("expected '=' after attribute " + key);
            ++i_;
            skip_space();
            if (eof() || (s_[i_] != '"' && s_[i_] != '\'')) fail("expected quoted value for attribute " + key);
            char quote = s_[i_++];
            size_t b = i_;
            while (!eof() && s_[i_] != quote) ++i_;
            if (eof()) fail("unterminated attribute value");
            el->attrs.emplace_back(std::move(key), decode(std::string(s_ + b, i_ - b)));
            ++i_;
        }
        // content
        for (;;) {
            if (eof()) fail("unterminated element <" + el->tag + ">");
            if (s_[i_] != '<') { ++i_; continue; }  // text: ignored
            if (starts("<!--")) { i_ += 4; skip_until("-->"); continue; }
            if (starts("<![CDATA[")) { i_ += 9; skip_until("]]>"); continue; }
            if (starts("<?")) { i_ += 2; skip_until("?>"); continue; }
            if (starts("</")) {
                i_ += 2;
                std::string name = parse_name();
                if (name != el->tag) fail("mismatched end tag </" + name + "> for <" + el->tag + ">");
                skip_space();
                if (eof() || s_[i_] != '>') fail("malformed end tag");
                ++i_;
                return el;
            }
            el->children.push_back(parse_element());
        }
    }
};

}  // namespace xml
}  // namespace ikgpu
