// Pattern compiler: PCRE-subset text -> syntax tree -> position automaton with boundary-condition
// nodes -> flat tables (hg_db.h), plus required-literal ("factor") analysis that feeds the
// streaming prefilter.  Host C++; replaces hs_compile_multi for the path at
// hypergrep/lib/c/hyperscanner.c:126-142,154-167.
//
// Accept / reject frontier follows Hyperscan 5.4's documented "unsupported constructs" (no
// look-around, back-references, atomic groups, possessive quantifiers, conditionals, recursion,
// callouts, \G \K \X \R \C, unicode properties without UCP), rejects expressions that can match the
// empty string (no HS_FLAG_ALLOWEMPTY), embedded start/end anchors outside multiline mode, and flag
// bits other than CASELESS|DOTALL|MULTILINE|SINGLEMATCH.  Byte semantics throughout (no UTF-8 mode).
#include "hg_compile.h"
#include "hg_core.h"

#include <algorithm>
#include <array>
#include <bitset>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <unordered_map>

namespace {

using ByteSet = std::bitset<256>;

struct CompileError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ---------------------------------------------------------------- truth tables for assertions
uint32_t tt_from(bool (*pred)(uint32_t pc, uint32_t nc)) {
  uint32_t t = 0;
  for (uint32_t pc = 0; pc < 4; pc++)
    for (uint32_t nc = 0; nc < 5; nc++)
      if (pred(pc, nc)) t |= 1u << (pc * 5 + nc);
  return t;
}
const uint32_t TT_BOL_ML = tt_from([](uint32_t pc, uint32_t) { return pc == HG_PC_START || pc == HG_PC_NL; });
const uint32_t TT_BOL = tt_from([](uint32_t pc, uint32_t) { return pc == HG_PC_START; });
const uint32_t TT_EOL_ML = tt_from([](uint32_t, uint32_t nc) { return nc == HG_NC_NL || nc == HG_NC_NLFINAL || nc == HG_NC_END; });
const uint32_t TT_EOL = tt_from([](uint32_t, uint32_t nc) { return nc == HG_NC_NLFINAL || nc == HG_NC_END; });
const uint32_t TT_EOD = tt_from([](uint32_t, uint32_t nc) { return nc == HG_NC_END; });
const uint32_t TT_WB = tt_from([](uint32_t pc, uint32_t nc) { return (pc == HG_PC_WORD) != (nc == HG_NC_WORD); });
const uint32_t TT_NWB = HG_TT_ALL & ~TT_WB;

// ---------------------------------------------------------------- syntax tree
struct Node {
  enum Kind { Empty, Class, Cat, Alt, Rep, Assert } kind = Empty;
  ByteSet cls;             // Class
  uint32_t tt = 0;         // Assert: truth table
  bool start_anchor = false, end_anchor = false;  // Assert: non-multiline ^ \A / $ \Z \z
  int min = 0, max = 0;    // Rep (max < 0: unbounded)
  std::vector<std::unique_ptr<Node>> kids;
};
using NodeP = std::unique_ptr<Node>;

NodeP make(Node::Kind k) {
  auto n = std::make_unique<Node>();
  n->kind = k;
  return n;
}

void add_caseless(ByteSet &s) {
  for (int c = 'a'; c <= 'z'; c++) {
    if (s[c]) s.set(c - 32);
    if (s[c - 32]) s.set(c);
  }
}
void set_range(ByteSet &s, int lo, int hi) {
  for (int c = lo; c <= hi; c++) s.set(c);
}
ByteSet cls_digit() { ByteSet s; set_range(s, '0', '9'); return s; }
ByteSet cls_word() { ByteSet s; set_range(s, '0', '9'); set_range(s, 'a', 'z'); set_range(s, 'A', 'Z'); s.set('_'); return s; }
ByteSet cls_space() { ByteSet s; s.set(' '); set_range(s, 9, 13); return s; }
ByteSet cls_hspace() { ByteSet s; s.set(9); s.set(' '); s.set(0xA0); return s; }
ByteSet cls_vspace() { ByteSet s; set_range(s, 10, 13); s.set(0x85); return s; }

// ---------------------------------------------------------------- parser
class Parser {
 public:
  Parser(const std::string &text, uint32_t flags) : s_(text) {
    caseless_ = flags & HG_FLAG_CASELESS;
    dotall_ = flags & HG_FLAG_DOTALL;
    multiline_ = flags & HG_FLAG_MULTILINE;
  }

  NodeP parse() {
    NodeP root = alternation();
    if (pos_ < s_.size()) throw CompileError("unmatched closing parenthesis");
    return root;
  }

 private:
  const std::string &s_;
  size_t pos_ = 0;
  bool caseless_ = false, dotall_ = false, multiline_ = false, extended_ = false;
  int depth_ = 0;

  bool eof() const { return pos_ >= s_.size(); }
  unsigned char peek(size_t k = 0) const { return static_cast<unsigned char>(s_[pos_ + k]); }
  bool has(size_t k) const { return pos_ + k < s_.size(); }

  struct FlagState { bool i, s, m, x; };
  FlagState save() const { return {caseless_, dotall_, multiline_, extended_}; }
  void restore(const FlagState &f) { caseless_ = f.i; dotall_ = f.s; multiline_ = f.m; extended_ = f.x; }

  NodeP byte_node(int b) const {
    NodeP n = make(Node::Class);
    n->cls.set(b);
    if (caseless_) add_caseless(n->cls);
    return n;
  }

  void skip_free_spacing() {
    if (!extended_) return;
    for (;;) {
      while (!eof() && (peek() == ' ' || (peek() >= 9 && peek() <= 13))) pos_++;
      if (!eof() && peek() == '#') {
        while (!eof() && peek() != '\n') pos_++;
        continue;
      }
      return;
    }
  }

  NodeP alternation() {
    NodeP alt = make(Node::Alt);
    for (;;) {
      alt->kids.push_back(sequence());
      if (!eof() && peek() == '|') { pos_++; continue; }
      break;
    }
    return alt;
  }

  NodeP sequence() {
    NodeP cat = make(Node::Cat);
    for (;;) {
      skip_free_spacing();
      if (eof() || peek() == '|' || peek() == ')') break;
      if (peek() == '*' || peek() == '+' || peek() == '?') throw CompileError("quantifier with nothing to repeat");
      std::vector<NodeP> atoms = atom();
      if (atoms.empty()) continue;
      // a quantifier binds to the last produced atom only (matters for \Q..\E runs)
      NodeP last = std::move(atoms.back());
      atoms.pop_back();
      for (auto &a : atoms) cat->kids.push_back(std::move(a));
      for (;;) {
        skip_free_spacing();
        if (eof()) break;
        int mn, mx;
        unsigned char q = peek();
        if (q == '*') { mn = 0; mx = -1; pos_++; }
        else if (q == '+') { mn = 1; mx = -1; pos_++; }
        else if (q == '?') { mn = 0; mx = 1; pos_++; }
        else if (q == '{') { if (!braces(mn, mx)) break; }
        else break;
        if (mx >= 0 && mx < mn) throw CompileError("numbers out of order in {} quantifier");
        if (mn > 32767 || mx > 32767) throw CompileError("bounded repeat is too large");
        if (!eof() && peek() == '+') throw CompileError("possessive quantifiers are not supported");
        if (!eof() && peek() == '?') pos_++;  // lazy: identical set of match end offsets
        if (last->kind == Node::Assert) throw CompileError("quantifier on a zero-width assertion");
        NodeP rep = make(Node::Rep);
        rep->min = mn;
        rep->max = mx;
        rep->kids.push_back(std::move(last));
        last = std::move(rep);
      }
      cat->kids.push_back(std::move(last));
    }
    return cat;
  }

  bool braces(int &mn, int &mx) {
    size_t q = pos_ + 1;
    auto number = [&](long &v) {
      int nd = 0;
      v = 0;
      while (q < s_.size() && s_[q] >= '0' && s_[q] <= '9') {
        v = std::min<long>(v * 10 + (s_[q] - '0'), 100000);
        q++;
        nd++;
      }
      return nd;
    };
    long a, b;
    if (!number(a)) return false;
    if (q < s_.size() && s_[q] == '}') { b = a; q++; }
    else if (q < s_.size() && s_[q] == ',') {
      q++;
      int nd = number(b);
      if (q >= s_.size() || s_[q] != '}') return false;
      q++;
      if (!nd) b = -1;
    } else return false;
    pos_ = q;
    mn = static_cast<int>(a);
    mx = static_cast<int>(b);
    return true;
  }

  static int hexval(int c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
  }

  // After a backslash: escapes denoting one byte. Returns -1 (position unchanged) if not such an escape.
  int byte_escape(bool in_class) {
    if (eof()) throw CompileError("pattern ends with a backslash");
    unsigned char c = peek();
    switch (c) {
      case 'n': pos_++; return '\n';
      case 'r': pos_++; return '\r';
      case 't': pos_++; return '\t';
      case 'f': pos_++; return '\f';
      case 'a': pos_++; return 7;
      case 'e': pos_++; return 27;
      case 'b': if (in_class) { pos_++; return 8; } return -1;
      case 'c': {
        if (!has(1)) throw CompileError("\\c at end of pattern");
        int x = peek(1);
        if (x >= 'a' && x <= 'z') x -= 32;
        pos_ += 2;
        return x ^ 0x40;
      }
      case 'x': {
        size_t q = pos_ + 1;
        unsigned v = 0;
        if (q < s_.size() && s_[q] == '{') {
          q++;
          int nd = 0;
          while (q < s_.size() && hexval(static_cast<unsigned char>(s_[q])) >= 0) {
            v = v * 16 + hexval(static_cast<unsigned char>(s_[q]));
            if (v > 0xFF) throw CompileError("\\x{} value does not fit a byte (UTF-8 mode is not supported)");
            q++;
            nd++;
          }
          if (q >= s_.size() || s_[q] != '}' || !nd) throw CompileError("malformed \\x{}");
          pos_ = q + 1;
          return static_cast<int>(v);
        }
        int nd = 0;
        while (nd < 2 && q < s_.size() && hexval(static_cast<unsigned char>(s_[q])) >= 0) {
          v = v * 16 + hexval(static_cast<unsigned char>(s_[q]));
          q++;
          nd++;
        }
        pos_ = q;
        return static_cast<int>(v);
      }
      default: break;
    }
    if (c >= '0' && c <= '7') {
      bool octal = c == '0' || in_class ||
                   (has(2) && peek(1) >= '0' && peek(1) <= '7' && peek(2) >= '0' && peek(2) <= '7');
      if (!octal) return -1;
      unsigned v = 0;
      int nd = 0;
      while (nd < 3 && !eof() && peek() >= '0' && peek() <= '7') {
        v = v * 8 + (peek() - '0');
        pos_++;
        nd++;
      }
      if (v > 0xFF) throw CompileError("octal escape does not fit a byte");
      return static_cast<int>(v);
    }
    bool alnum = (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z');
    if (!alnum) { pos_++; return c; }
    return -1;
  }

  bool class_escape(ByteSet &out) {
    if (eof()) return false;
    ByteSet s;
    bool neg = false;
    switch (peek()) {
      case 'd': s = cls_digit(); break;
      case 'D': s = cls_digit(); neg = true; break;
      case 'w': s = cls_word(); break;
      case 'W': s = cls_word(); neg = true; break;
      case 's': s = cls_space(); break;
      case 'S': s = cls_space(); neg = true; break;
      case 'h': s = cls_hspace(); break;
      case 'H': s = cls_hspace(); neg = true; break;
      case 'v': s = cls_vspace(); break;
      case 'V': s = cls_vspace(); neg = true; break;
      case 'N': s.set('\n'); neg = true; break;
      default: return false;
    }
    pos_++;
    out |= neg ? ~s : s;
    return true;
  }

  static bool posix_set(const std::string &name, ByteSet &s) {
    if (name == "alpha") { set_range(s, 'a', 'z'); set_range(s, 'A', 'Z'); }
    else if (name == "digit") s |= cls_digit();
    else if (name == "alnum") { set_range(s, 'a', 'z'); set_range(s, 'A', 'Z'); s |= cls_digit(); }
    else if (name == "upper") set_range(s, 'A', 'Z');
    else if (name == "lower") set_range(s, 'a', 'z');
    else if (name == "space") s |= cls_space();
    else if (name == "blank") { s.set(' '); s.set(9); }
    else if (name == "punct") { for (int b = 33; b < 127; b++) if (!hg_is_word(b) || b == '_') s.set(b); }
    else if (name == "print") set_range(s, 32, 126);
    else if (name == "graph") set_range(s, 33, 126);
    else if (name == "cntrl") { set_range(s, 0, 31); s.set(127); }
    else if (name == "xdigit") { s |= cls_digit(); set_range(s, 'a', 'f'); set_range(s, 'A', 'F'); }
    else if (name == "word") s |= cls_word();
    else if (name == "ascii") set_range(s, 0, 127);
    else return false;
    return true;
  }

  NodeP bracket() {  // pos_ just past '['
    ByteSet set;
    bool neg = false;
    if (!eof() && peek() == '^') { neg = true; pos_++; }
    bool first = true;
    for (;;) {
      if (eof()) throw CompileError("unterminated character class");
      unsigned char c = peek();
      if (c == ']' && !first) { pos_++; break; }
      first = false;
      int lo;
      if (c == '[' && has(1) && peek(1) == ':') {
        size_t q = pos_ + 2;
        bool pneg = false;
        if (q < s_.size() && s_[q] == '^') { pneg = true; q++; }
        size_t e = s_.find(":]", q);
        if (e != std::string::npos) {
          ByteSet ps;
          if (!posix_set(s_.substr(q, e - q), ps)) throw CompileError("unknown POSIX class name");
          set |= pneg ? ~ps : ps;
          pos_ = e + 2;
          continue;
        }
      }
      if (c == '\\') {
        pos_++;
        if (eof()) throw CompileError("pattern ends inside a character class");
        if (class_escape(set)) continue;
        lo = byte_escape(true);
        if (lo < 0) throw CompileError("unsupported escape in character class");
      } else {
        lo = c;
        pos_++;
      }
      if (has(1) && peek() == '-' && peek(1) != ']') {
        pos_++;
        int hi;
        if (peek() == '\\') {
          pos_++;
          ByteSet tmp;
          if (class_escape(tmp)) {  // "a-\d": the '-' is a literal
            set.set(lo);
            set.set('-');
            set |= tmp;
            continue;
          }
          hi = byte_escape(true);
          if (hi < 0) throw CompileError("unsupported escape in character class range");
        } else if (peek() == '[' && has(1) && peek(1) == ':') {
          throw CompileError("POSIX class used as a range endpoint");
        } else {
          hi = peek();
          pos_++;
        }
        if (hi < lo) throw CompileError("range out of order in character class");
        set_range(set, lo, hi);
      } else {
        set.set(lo);
      }
    }
    if (caseless_) add_caseless(set);
    if (neg) set = ~set;
    if (set.none()) throw CompileError("character class matches nothing");
    NodeP n = make(Node::Class);
    n->cls = set;
    return n;
  }

  NodeP assertion(uint32_t tt, bool start_anchor = false, bool end_anchor = false) {
    NodeP n = make(Node::Assert);
    n->tt = tt;
    n->start_anchor = start_anchor;
    n->end_anchor = end_anchor;
    return n;
  }

  NodeP close_group(NodeP inner) {
    if (eof() || peek() != ')') throw CompileError("missing closing parenthesis");
    pos_++;
    return inner;
  }

  // Returns zero or more atoms (zero: option-setting group or comment; several: a \Q..\E run).
  std::vector<NodeP> atom() {
    std::vector<NodeP> out;
    unsigned char c = peek();
    if (c == '(') {
      pos_++;
      if (++depth_ > 200) throw CompileError("pattern nesting too deep");
      if (!eof() && peek() == '*') throw CompileError("backtracking control verbs are not supported");
      if (!eof() && peek() == '?') {
        pos_++;
        group_extension(out);
      } else {
        FlagState f = save();
        NodeP inner = alternation();
        restore(f);
        out.push_back(close_group(std::move(inner)));
      }
      depth_--;
      return out;
    }
    if (c == '[') { pos_++; out.push_back(bracket()); return out; }
    if (c == '.') {
      pos_++;
      NodeP n = make(Node::Class);
      n->cls.set();
      if (!dotall_) n->cls.reset('\n');
      out.push_back(std::move(n));
      return out;
    }
    if (c == '^') { pos_++; out.push_back(multiline_ ? assertion(TT_BOL_ML) : assertion(TT_BOL, true, false)); return out; }
    if (c == '$') { pos_++; out.push_back(multiline_ ? assertion(TT_EOL_ML) : assertion(TT_EOL, false, true)); return out; }
    if (c != '\\') { pos_++; out.push_back(byte_node(c)); return out; }
    pos_++;
    if (eof()) throw CompileError("pattern ends with a backslash");
    unsigned char e = peek();
    switch (e) {
      case 'b': pos_++; out.push_back(assertion(TT_WB)); return out;
      case 'B': pos_++; out.push_back(assertion(TT_NWB)); return out;
      case 'A': pos_++; out.push_back(assertion(TT_BOL, true, false)); return out;
      case 'Z': pos_++; out.push_back(assertion(TT_EOL, false, true)); return out;
      case 'z': pos_++; out.push_back(assertion(TT_EOD, false, true)); return out;
      case 'E': pos_++; return out;
      case 'Q': {
        pos_++;
        while (!eof()) {
          if (has(1) && peek() == '\\' && peek(1) == 'E') { pos_ += 2; break; }
          out.push_back(byte_node(peek()));
          pos_++;
        }
        return out;
      }
      case 'G': case 'K': case 'X': case 'R': case 'C':
        throw CompileError(std::string("\\") + static_cast<char>(e) + " is not supported");
      case 'p': case 'P': throw CompileError("unicode properties need UCP mode, which is not supported");
      case 'g': case 'k': throw CompileError("back-references are not supported");
      default: break;
    }
    ByteSet cs;
    if (class_escape(cs)) {
      NodeP n = make(Node::Class);
      n->cls = cs;
      if (caseless_) add_caseless(n->cls);
      out.push_back(std::move(n));
      return out;
    }
    int b = byte_escape(false);
    if (b < 0) {
      if (e >= '1' && e <= '9') throw CompileError("back-references are not supported");
      throw CompileError(std::string("unsupported escape \\") + static_cast<char>(e));
    }
    out.push_back(byte_node(b));
    return out;
  }

  void group_extension(std::vector<NodeP> &out) {  // pos_ just past "(?"
    if (eof()) throw CompileError("unterminated group");
    unsigned char c = peek();
    auto scoped = [&]() {
      FlagState f = save();
      NodeP inner = alternation();
      restore(f);
      out.push_back(close_group(std::move(inner)));
    };
    if (c == '#') {
      size_t e = s_.find(')', pos_);
      if (e == std::string::npos) throw CompileError("unterminated comment");
      pos_ = e + 1;
      return;
    }
    if (c == ':') { pos_++; scoped(); return; }
    if (c == '=' || c == '!') throw CompileError("lookahead assertions are not supported");
    if (c == '>') throw CompileError("atomic groups are not supported");
    if (c == '(') throw CompileError("conditional subpatterns are not supported");
    if (c == 'R' || c == '&' || c == '+' || (c >= '0' && c <= '9')) throw CompileError("recursion and subroutine calls are not supported");
    if (c == 'C') throw CompileError("callouts are not supported");
    if (c == '|') throw CompileError("branch reset groups are not supported");
    if (c == '<' || c == 'P' || c == '\'') {
      size_t q = pos_;
      char close = '>';
      if (c == 'P') {
        q++;
        if (q < s_.size() && (s_[q] == '=' || s_[q] == '>')) throw CompileError("named back-references and recursion are not supported");
        if (q >= s_.size() || s_[q] != '<') throw CompileError("malformed (?P group");
        q++;
      } else if (c == '<') {
        q++;
        if (q < s_.size() && (s_[q] == '=' || s_[q] == '!')) throw CompileError("lookbehind assertions are not supported");
      } else {
        q++;
        close = '\'';
      }
      size_t start = q;
      while (q < s_.size() && hg_is_word(static_cast<unsigned char>(s_[q]))) q++;
      if (q == start || q >= s_.size() || s_[q] != close) throw CompileError("malformed group name");
      pos_ = q + 1;
      scoped();
      return;
    }
    // inline options
    bool on = true, any = false;
    FlagState f = save();
    while (!eof()) {
      c = peek();
      bool *target = nullptr;
      if (c == 'i') target = &f.i;
      else if (c == 's') target = &f.s;
      else if (c == 'm') target = &f.m;
      else if (c == 'x') target = &f.x;
      else if (c == '-') { on = false; any = true; pos_++; continue; }
      else break;
      *target = on;
      any = true;
      pos_++;
    }
    if (!any || eof()) throw CompileError("unsupported group construct");
    if (peek() == ')') {  // applies to the rest of the enclosing group
      pos_++;
      restore(f);
      return;
    }
    if (peek() == ':') {
      pos_++;
      FlagState outer = save();
      restore(f);
      NodeP inner = alternation();
      restore(outer);
      out.push_back(close_group(std::move(inner)));
      return;
    }
    throw CompileError("unsupported inline option");
  }
};

// ---------------------------------------------------------------- static checks
bool can_consume(const Node &n) {
  switch (n.kind) {
    case Node::Class: return true;
    case Node::Cat: case Node::Alt:
      for (auto &k : n.kids) if (can_consume(*k)) return true;
      return false;
    case Node::Rep: return n.max != 0 && can_consume(*n.kids[0]);
    default: return false;
  }
}

void check_embedded_anchors(const Node &n, bool before, bool after) {
  switch (n.kind) {
    case Node::Assert:
      if (n.start_anchor && before) throw CompileError("embedded start anchors are not supported");
      if (n.end_anchor && after) throw CompileError("embedded end anchors are not supported");
      break;
    case Node::Cat:
      for (size_t i = 0; i < n.kids.size(); i++) {
        bool b = before, a = after;
        for (size_t j = 0; j < i; j++) b = b || can_consume(*n.kids[j]);
        for (size_t j = i + 1; j < n.kids.size(); j++) a = a || can_consume(*n.kids[j]);
        check_embedded_anchors(*n.kids[i], b, a);
      }
      break;
    case Node::Alt:
      for (auto &k : n.kids) check_embedded_anchors(*k, before, after);
      break;
    case Node::Rep: {
      bool loops = (n.max < 0 || n.max > 1) && can_consume(*n.kids[0]);
      check_embedded_anchors(*n.kids[0], before || loops, after || loops);
      break;
    }
    default: break;
  }
}

// Longest string the expression can match; -1 = unbounded (or absurdly long).
long max_match_len(const Node &n) {
  const long kCap = 1 << 20;
  switch (n.kind) {
    case Node::Empty:
    case Node::Assert: return 0;
    case Node::Class: return 1;
    case Node::Cat: {
      long t = 0;
      for (auto &k : n.kids) {
        long v = max_match_len(*k);
        if (v < 0) return -1;
        t += v;
        if (t > kCap) return -1;
      }
      return t;
    }
    case Node::Alt: {
      long t = 0;
      for (auto &k : n.kids) {
        long v = max_match_len(*k);
        if (v < 0) return -1;
        t = std::max(t, v);
      }
      return t;
    }
    case Node::Rep: {
      long v = n.kids.empty() ? 0 : max_match_len(*n.kids[0]);
      if (v < 0) return -1;
      if (v == 0) return 0;
      if (n.max < 0) return -1;
      long t = v * n.max;
      return t > kCap ? -1 : t;
    }
  }
  return -1;
}

bool has_assert(const Node &n) {
  if (n.kind == Node::Assert) return true;
  for (auto &k : n.kids) if (has_assert(*k)) return true;
  return false;
}

// ---------------------------------------------------------------- position automaton
// Contexts in which the expression matches the empty string (a 20-bit truth table, HG_TT_ALL = always).
uint32_t nullable_tt(const Node &n) {
  switch (n.kind) {
    case Node::Empty: return HG_TT_ALL;
    case Node::Assert: return n.tt;
    case Node::Class: return 0;
    case Node::Cat: {
      uint32_t t = HG_TT_ALL;
      for (auto &k : n.kids) t &= nullable_tt(*k);
      return t;
    }
    case Node::Alt: {
      uint32_t t = 0;
      for (auto &k : n.kids) t |= nullable_tt(*k);
      return t;
    }
    case Node::Rep: return (n.min == 0 || n.kids.empty()) ? HG_TT_ALL : nullable_tt(*n.kids[0]);
  }
  return 0;
}

// Size of the expression as a Thompson program (one instruction per class / assertion, two per alternation branch point,
// a split per optional copy, split + jump per loop): the measure Hyperscan's graph limits and the oracle (orx.c gen())
// bound an expression by.  Saturates.
uint64_t program_size(const Node &n) {
  const uint64_t kCap = uint64_t(1) << 40;
  switch (n.kind) {
    case Node::Empty: return 0;
    case Node::Class: case Node::Assert: return 1;
    case Node::Cat: {
      uint64_t t = 0;
      for (auto &k : n.kids) t = std::min(kCap, t + program_size(*k));
      return t;
    }
    case Node::Alt: {
      if (n.kids.size() == 1) return program_size(*n.kids[0]);
      uint64_t t = 2 * (n.kids.size() - 1);
      for (auto &k : n.kids) t = std::min(kCap, t + program_size(*k));
      return t;
    }
    case Node::Rep: {
      const uint64_t k = program_size(*n.kids[0]);
      const uint64_t opt = n.max < 0 ? k + 2 : static_cast<uint64_t>(n.max - n.min) * (k + 1);
      return std::min(kCap, static_cast<uint64_t>(n.min) * k + opt);
    }
  }
  return 0;
}

struct Cond { uint32_t pos, tt; };
struct Frag {
  uint32_t nullable = 0;  // contexts in which the fragment matches the empty string
  std::vector<Cond> first, last;
};
struct Edge { uint32_t p, q, tt; };

// Fragments own their position sets and are consumed (moved) by the combinators: every position belongs to exactly one
// fragment at a time, so first / last sets only ever grow by appending, and an expression of n positions costs O(n + edges)
// whatever its repeat counts (a{32767}: 32767 positions, 32766 edges).
class Glushkov {
 public:
  std::vector<ByteSet> pos_class;
  std::vector<Edge> edges;  // (p, q, tt), possibly repeated: the table builder merges them

  Frag build(const Node &n) {
    switch (n.kind) {
      case Node::Empty: { Frag f; f.nullable = HG_TT_ALL; return f; }
      case Node::Assert: { Frag f; f.nullable = n.tt; return f; }
      case Node::Class: {
        if (pos_class.size() >= HG_HUGE_MAX_NODES) throw CompileError("pattern too large");
        uint32_t p = static_cast<uint32_t>(pos_class.size());
        pos_class.push_back(n.cls);
        Frag f;
        f.first.push_back({p, HG_TT_ALL});
        f.last.push_back({p, HG_TT_ALL});
        return f;
      }
      case Node::Cat: {
        Frag acc;
        acc.nullable = HG_TT_ALL;
        for (auto &k : n.kids) acc = cat(std::move(acc), build(*k));
        return acc;
      }
      case Node::Alt: {
        Frag acc = build(*n.kids[0]);
        for (size_t i = 1; i < n.kids.size(); i++) acc = alt(std::move(acc), build(*n.kids[i]));
        return acc;
      }
      case Node::Rep: {
        const Node &k = *n.kids[0];
        // A body that can ALWAYS match the empty string: k{n,m} = (k+){0,m} and k{n,} = (k+)*, where k+ is k without the
        // empty match — the same positions and edges with the nullable flag cleared.  Copies then link to their neighbours
        // only, instead of every copy to every later one ((a?){n} would have n^2 / 2 edges).
        const bool strip = nullable_tt(k) == HG_TT_ALL;
        auto copy = [&]() {
          Frag f = build(k);
          if (strip) f.nullable = 0;
          return f;
        };
        const int mn = strip ? 0 : n.min;
        Frag acc;
        acc.nullable = HG_TT_ALL;
        for (int i = 0; i < mn; i++) acc = cat(std::move(acc), copy());
        if (n.max < 0) {
          Frag loop = copy();
          link(loop.last, loop.first);
          loop.nullable = HG_TT_ALL;
          acc = cat(std::move(acc), std::move(loop));
        } else if (n.max > mn) {
          // k{0,m} as (((k)?k)?k)? : a string of j copies uses the LAST j of them, so the optional part is entered at any
          // copy (the entering node's range of targets) and left from the last one only; copies are laid out in the order
          // they are matched, which makes the edge from one copy to the next a node -> node + 1 edge wherever k is a chain.
          Frag opt;
          opt.nullable = HG_TT_ALL;
          for (int i = mn; i < n.max; i++) {
            opt = cat(std::move(opt), copy());
            opt.nullable = HG_TT_ALL;
          }
          acc = cat(std::move(acc), std::move(opt));
        }
        return acc;
      }
    }
    return Frag();
  }

 private:
  void link(const std::vector<Cond> &from, const std::vector<Cond> &to) {
    if (edges.size() + from.size() * to.size() > HG_HUGE_MAX_EDGES) throw CompileError("pattern too large");
    for (auto &l : from)
      for (auto &f : to) {
        uint32_t tt = l.tt & f.tt;
        if (tt) edges.push_back({l.pos, f.pos, tt});
      }
  }
  Frag cat(Frag a, Frag b) {
    link(a.last, b.first);
    Frag r;
    r.nullable = a.nullable & b.nullable;
    r.first = std::move(a.first);
    if (a.nullable)
      for (auto &f : b.first)
        if (f.tt & a.nullable) r.first.push_back({f.pos, f.tt & a.nullable});
    r.last = std::move(b.last);
    if (b.nullable)
      for (auto &l : a.last)
        if (l.tt & b.nullable) r.last.push_back({l.pos, l.tt & b.nullable});
    return r;
  }
  static Frag alt(Frag a, Frag b) {
    Frag r = std::move(a);
    r.nullable |= b.nullable;
    r.first.insert(r.first.end(), b.first.begin(), b.first.end());
    r.last.insert(r.last.end(), b.last.begin(), b.last.end());
    return r;
  }
};

// ---------------------------------------------------------------- required-literal analysis
struct Lit {
  std::string bytes, cmask;  // cmask: 0xFF exact, 0xDF either case
  bool operator<(const Lit &o) const { return std::tie(bytes, cmask) < std::tie(o.bytes, o.cmask); }
  bool operator==(const Lit &o) const { return bytes == o.bytes && cmask == o.cmask; }
};
using LitSet = std::vector<Lit>;
constexpr size_t MAX_EXACT = 16;      // strings in an exact set / cover
constexpr size_t MAX_EXACT_LEN = 96;  // bytes per string while concatenating

struct Info {
  bool exact = false;  // language of the node is exactly `set`
  LitSet set;
  bool has_cover = false;  // every match contains one of `cover`
  LitSet cover;
  long lead = 0;  // ... beginning at most this many bytes after the match's start (-1: no bound); the confirm window's size
};
long add_len(long a, long b) {  // lengths with -1 = unbounded
  if (a < 0 || b < 0 || a + b > (1 << 20)) return -1;
  return a + b;
}

int byte_commonness(unsigned char b) {
  if (b == ' ') return 10;
  if (b >= 'a' && b <= 'z') return std::strchr("etaoinsrhl", b) ? 8 : 5;
  if (b >= '0' && b <= '9') return 6;
  if (std::strchr("=:-./_", b)) return 4;
  if (b >= 'A' && b <= 'Z') return 2;
  if (b > 32 && b < 127) return 2;
  return 1;
}
size_t min_len(const LitSet &s) {
  size_t m = SIZE_MAX;
  for (auto &l : s) m = std::min(m, l.bytes.size());
  return s.empty() ? 0 : m;
}
void dedupe(LitSet &s) {
  std::sort(s.begin(), s.end());
  s.erase(std::unique(s.begin(), s.end()), s.end());
}
// Better cover: longer shortest literal (saturating at 16), then fewer literals, then longer.
bool better_cover(const LitSet &a, const LitSet &b) {
  size_t ma = std::min<size_t>(min_len(a), 16), mb = std::min<size_t>(min_len(b), 16);
  if (ma != mb) return ma > mb;
  if (a.size() != b.size()) return a.size() < b.size();
  return min_len(a) > min_len(b);
}
void offer(Info &info, const LitSet &cand, long lead) {
  if (cand.empty() || min_len(cand) == 0 || cand.size() > MAX_EXACT) return;
  if (!info.has_cover || better_cover(cand, info.cover)) {
    info.has_cover = true;
    info.cover = cand;
    info.lead = lead;
  }
}
bool cross(const LitSet &a, const LitSet &b, LitSet &out) {
  if (a.size() * b.size() > MAX_EXACT) return false;
  out.clear();
  for (auto &x : a)
    for (auto &y : b) {
      if (x.bytes.size() + y.bytes.size() > MAX_EXACT_LEN) return false;
      out.push_back({x.bytes + y.bytes, x.cmask + y.cmask});
    }
  dedupe(out);
  return true;
}

Info analyze(const Node &n) {
  Info r;
  switch (n.kind) {
    case Node::Empty: case Node::Assert:
      r.exact = true;
      r.set.push_back({"", ""});
      return r;
    case Node::Class: {
      size_t cnt = n.cls.count();
      ByteSet folded;
      for (int b = 0; b < 256; b++) if (n.cls[b]) folded.set((b >= 'A' && b <= 'Z') ? b + 32 : b);
      bool case_closed = true;
      for (int b = 'a'; b <= 'z'; b++) if (n.cls[b] != n.cls[b - 32]) case_closed = false;
      if (cnt <= 10 || (case_closed && folded.count() <= 10)) {
        r.exact = true;
        for (int b = 0; b < 256; b++) {
          if (!n.cls[b]) continue;
          if (b >= 'A' && b <= 'Z' && n.cls[b + 32]) continue;  // represented by the lower-case entry
          bool both = b >= 'a' && b <= 'z' && n.cls[b - 32];
          r.set.push_back({std::string(1, static_cast<char>(b)), std::string(1, static_cast<char>(both ? 0xDF : 0xFF))});
        }
      }
      return r;
    }
    case Node::Cat: {
      LitSet run{{"", ""}};
      bool all_exact = true;
      long before = 0;    // the most bytes the kids before the current one can match
      long run_lead = 0;  // ... before the first kid of the current run of exact kids
      auto flush = [&]() {
        offer(r, run, run_lead);
        run = LitSet{{"", ""}};
      };
      for (auto &k : n.kids) {
        Info ki = analyze(*k);
        const long after = add_len(before, max_match_len(*k));
        if (ki.has_cover) offer(r, ki.cover, add_len(before, ki.lead));
        if (ki.exact) {
          LitSet next;
          if (cross(run, ki.set, next)) run = std::move(next);
          else {
            all_exact = false;
            flush();
            run = ki.set;
            run_lead = before;  // the new run begins with this kid
            if (run.size() > MAX_EXACT) { run = LitSet{{"", ""}}; run_lead = after; }
          }
        } else {
          all_exact = false;
          flush();
          run_lead = after;  // the next run begins behind this kid
        }
        before = after;
      }
      if (all_exact) { r.exact = true; r.set = run; }
      offer(r, run, run_lead);
      return r;
    }
    case Node::Alt: {
      bool all_exact = true, all_cover = true;
      LitSet uni, cov;
      long cov_lead = 0;  // the largest lead among the branches' covers (a branch's whole literal set begins with the match: 0)
      for (auto &k : n.kids) {
        Info ki = analyze(*k);
        if (ki.exact) { uni.insert(uni.end(), ki.set.begin(), ki.set.end()); }
        else all_exact = false;
        const LitSet *c = ki.has_cover ? &ki.cover : (ki.exact && min_len(ki.set) > 0 ? &ki.set : nullptr);
        if (ki.exact && ki.has_cover && min_len(ki.set) > 0 && better_cover(ki.set, ki.cover)) c = &ki.set;
        if (c) {
          cov.insert(cov.end(), c->begin(), c->end());
          const long l = c == &ki.cover ? ki.lead : 0;
          cov_lead = (cov_lead < 0 || l < 0) ? -1 : std::max(cov_lead, l);
        }
        else all_cover = false;
      }
      dedupe(uni);
      dedupe(cov);
      if (all_exact && uni.size() <= MAX_EXACT) { r.exact = true; r.set = uni; }
      if (all_cover) offer(r, cov, cov_lead);
      if (r.exact) offer(r, r.set, 0);
      return r;
    }
    case Node::Rep: {
      Info ki = analyze(*n.kids[0]);
      if (ki.exact && n.min == n.max && n.min <= 64) {
        LitSet acc{{"", ""}};
        bool ok = true;
        for (int i = 0; i < n.min && ok; i++) {
          LitSet next;
          ok = cross(acc, ki.set, next);
          if (ok) acc = std::move(next);
        }
        if (ok) { r.exact = true; r.set = acc; offer(r, acc, 0); return r; }
      }
      if (n.min >= 1) {
        if (ki.has_cover) offer(r, ki.cover, ki.lead);  // (the occurrence inside the FIRST copy)
        if (ki.exact) {  // at least `min` consecutive copies are required
          LitSet acc{{"", ""}};
          for (int i = 0; i < std::min(n.min, 32); i++) {
            LitSet next;
            if (!cross(acc, ki.set, next)) break;
            acc = std::move(next);
          }
          offer(r, acc, 0);
        }
      }
      return r;
    }
  }
  return r;
}

// Keep at most HG_FACTOR_MAX bytes of a long literal: the sub-run whose rarest window is rarest.
Lit clip(const Lit &l, size_t *offset = nullptr) {
  if (offset) *offset = 0;
  if (l.bytes.size() <= HG_FACTOR_MAX) return l;
  size_t best = 0;
  int best_score = INT32_MAX;
  for (size_t o = 0; o + HG_FACTOR_MAX <= l.bytes.size(); o++) {
    int s = 0;
    for (size_t i = 0; i < HG_FACTOR_MAX; i++) s += byte_commonness(static_cast<unsigned char>(l.bytes[o + i]));
    if (s < best_score) { best_score = s; best = o; }
  }
  if (offset) *offset = best;
  return {l.bytes.substr(best, HG_FACTOR_MAX), l.cmask.substr(best, HG_FACTOR_MAX)};
}

// Occurrence counts of dword-aligned windows in a text sample (hgc_tune).
struct SampleStats {
  std::unordered_map<uint32_t, uint32_t> c4;  // folded dword
  std::unordered_map<uint64_t, uint32_t> c6;  // folded dword | folded first two bytes of the next dword << 32
  size_t dwords = 0;
};

// (Re)build windows, bucket index, LDS filter slots and neighbour conditions from db.factors.
int build_filter(HgDb &db, const SampleStats *stats, std::string *err) {
  db.windows.clear();
  std::vector<std::pair<uint32_t, HgWindow>> keyed;
  std::vector<uint32_t> next16_of;  // per keyed entry: hg_next16 of the two bytes after the window, 0 if unknown
  const uint32_t fold = db.fold_mask;
  const uint32_t wbytes = db.window_bytes, wmask = db.window_mask;
  // contenders[v]: the literals (of different patterns or the same) that contain the window value v at some offset
  std::unordered_map<uint32_t, uint32_t> contenders;
  for (uint32_t fi = 0; fi < db.nreal_factors; fi++) {
    const HgFactor &fct = db.factors[fi];
    std::vector<uint32_t> mine;
    for (uint32_t o = 0; o + wbytes <= fct.len; o++) {
      uint32_t v = 0;
      std::memcpy(&v, fct.lit + o, wbytes);
      mine.push_back((v | fold) & wmask);
    }
    std::sort(mine.begin(), mine.end());
    mine.erase(std::unique(mine.begin(), mine.end()), mine.end());
    for (uint32_t v : mine) contenders[v]++;
  }
  // the case variants of the window at offset o of a literal: all of them when nothing is folded (fold == 0), else the one folded value
  auto variants = [&](const HgFactor &fct, uint32_t o, std::vector<uint32_t> &out) {
    uint32_t v = 0;
    std::memcpy(&v, fct.lit + o, wbytes);
    out.assign(1, (v | fold) & wmask);
    if (fold) return;
    for (uint32_t b = 0; b < wbytes; b++)
      if ((fct.casebits >> (o + b)) & 1u) {  // a case-insensitive letter (stored in lower case): both cases
        const size_t have = out.size();
        for (size_t i = 0; i < have; i++) out.push_back(out[i] ^ (0x20u << (8 * b)));
      }
  };
  std::vector<uint32_t> vars;
  for (uint32_t fi = 0; fi < db.nreal_factors; fi++) {
    const HgFactor &fct = db.factors[fi];
    Lit l{std::string(reinterpret_cast<const char *>(fct.lit), fct.len), std::string(fct.len, '\xFF')};
    for (uint32_t b = 0; b < fct.len; b++) l.cmask[b] = static_cast<char>(hg_factor_cmask(fct, b));
    if (db.dense && fct.len < wbytes) {
      // a literal one byte short of a window: every value of the byte after it (the stream pass reads zeros past the text)
      std::vector<uint32_t> seen;
      for (uint32_t last = 0; last < 256; last++) {
        uint32_t v = 0;
        std::memcpy(&v, fct.lit, fct.len);
        v = ((v | (last << 24)) | fold) & wmask;
        if (std::find(seen.begin(), seen.end(), v) != seen.end()) continue;  // folding maps two bytes onto one value
        seen.push_back(v);
        keyed.push_back({hg_hash_window(v), HgWindow{v, fi << 8}});
        next16_of.push_back(0);
      }
      continue;
    }
    // One window per residue mod 4 (the stream pass probes dword-aligned windows) — or, with byte-aligned probing
    // (db.dense), ONE window per literal at any offset.
    const uint32_t step = db.dense ? db.dense : 4u;  // the stream pass probes a window every `step` bytes: one window per residue mod step
    for (uint32_t res = 0; res < step; res++) {
      // The stream kernel compares the window in its hot path and the 12-byte neighbourhood [o-4, o+8) in the second level.
      // With sample statistics: take the offset whose window dword is rarest in the sample (exactly: one occurrence in a
      // 1 MiB sample is 30 000 false candidates in 32 GiB — config 5 went from 42 M to 64 M candidates when small counts
      // were treated as equal).  Then the offset whose window value no OTHER literal contains anywhere
      // (`contenders`: a window with one owner goes from the direct table straight to its literal; K%04x-%08x literals all
      // contain "xxx-" windows that sixteen of them share).  Then the offset whose known bytes are the most selective by a
      // static byte-frequency table.
      int best = -1;
      long best_cost = 0;
      uint32_t best_shared = 0;
      int best_sel = -1;
      for (uint32_t o = res; o + wbytes <= fct.len; o += step) {
        uint32_t v = 0;
        std::memcpy(&v, fct.lit + o, wbytes);
        v = (v | fold) & wmask;
        long cost = 0;
        if (stats) {  // the hot path compares the window dword alone: its frequency in the sample is what costs (all case variants)
          variants(fct, o, vars);
          for (uint32_t x : vars) {
            auto it = stats->c4.find(x);
            cost += it == stats->c4.end() ? 0 : it->second;
          }
        }
        const uint32_t shared = contenders[v] > 1 ? contenders[v] : 0;
        int sel = 0;
        for (int j = static_cast<int>(o); j < static_cast<int>(o + wbytes); j++) sel += 11 - byte_commonness(static_cast<unsigned char>(l.bytes[j]));
        for (int j = std::max(static_cast<int>(o) - 4, 0); j < std::min<int>(o + 7, fct.len); j++)
          sel += (11 - byte_commonness(static_cast<unsigned char>(l.bytes[j]))) / 4;
        if (best < 0 || cost < best_cost || (cost == best_cost && (shared < best_shared || (shared == best_shared && sel > best_sel)))) {
          best = static_cast<int>(o);
          best_cost = cost;
          best_shared = shared;
          best_sel = sel;
        }
      }
      if (best < 0) continue;  // cannot happen for len >= HG_FAST_MIN_FACTOR
      // (case-insensitive positions hold lower-case letters already: folding maps both cases onto them; without folding every
      // case variant of the window is a window of its own)
      variants(fct, static_cast<uint32_t>(best), vars);
      for (uint32_t x : vars) {
        keyed.push_back({hg_hash_window(x), HgWindow{x, (fi << 8) | static_cast<uint32_t>(best)}});
        next16_of.push_back(0);
      }
    }
  }
  {  // sort by bucket, carrying next16 along
    std::vector<size_t> order(keyed.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t x, size_t y) { return keyed[x].first < keyed[y].first; });
    std::vector<std::pair<uint32_t, HgWindow>> k2;
    std::vector<uint32_t> n2;
    for (size_t i : order) { k2.push_back(keyed[i]); n2.push_back(next16_of[i]); }
    keyed.swap(k2);
    next16_of.swap(n2);
  }
  db.bucket_off.assign((1u << HG_HASH_BITS) + 1, 0);
  for (auto &kw : keyed) {
    db.bucket_off[kw.first + 1]++;
    db.windows.push_back(kw.second);
  }
  for (size_t i = 1; i < db.bucket_off.size(); i++) db.bucket_off[i] += db.bucket_off[i - 1];
  if (db.windows.empty()) db.windows.push_back(HgWindow{0, 0});  // keep device arrays non-empty

  // Discriminated buckets for the GPU verify pass (hg_db.h): per hash-C group the literal dword that splits it best.
  {
    db.disc.assign(size_t(1) << HG_HASH_BITS, 0);
    std::vector<std::pair<uint32_t, HgWindow>> keyed2;
    for (size_t g0 = 0; g0 < keyed.size();) {
      size_t g1 = g0;
      while (g1 < keyed.size() && keyed[g1].first == keyed[g0].first) g1++;
      const uint32_t h = keyed[g0].first;
      int best_delta = 0;
      uint32_t best_sel = 0;
      size_t best_distinct = 1;
      auto key_of = [&](const HgWindow &w, int delta, uint32_t sel) {
        const HgFactor &fct = db.factors[w.factor_off >> 8];
        const int at = static_cast<int>(w.factor_off & 0xff) + delta;
        uint32_t v = 0;
        for (int b = 0; b < 4; b++)
          if ((sel >> b) & 1u) v |= static_cast<uint32_t>(fct.lit[at + b]) << (8 * b);
        return (v | fold) & hg_disc_bytes(sel);
      };
      if (g1 - g0 > 2) {
        for (int delta = -16; delta <= 28; delta += 4) {
          if (delta == 0) continue;
          uint32_t sel = 15;
          for (size_t e = g0; e < g1; e++) {
            const HgFactor &fct = db.factors[keyed[e].second.factor_off >> 8];
            const int at = static_cast<int>(keyed[e].second.factor_off & 0xff) + delta;
            for (int b = 0; b < 4; b++) {
              if (at + b < 0 || at + b >= static_cast<int>(fct.len)) sel &= ~(1u << b);
              else if (!fold && hg_factor_cmask(fct, at + b) != 0xFF) sel &= ~(1u << b);  // (nothing is folded: a case-insensitive letter discriminates nothing)
            }
          }
          if (!sel) continue;
          std::vector<uint32_t> ks;
          for (size_t e = g0; e < g1; e++) ks.push_back(key_of(keyed[e].second, delta, sel));
          std::sort(ks.begin(), ks.end());
          const size_t distinct = static_cast<size_t>(std::unique(ks.begin(), ks.end()) - ks.begin());
          if (distinct > best_distinct || (distinct == best_distinct && distinct > 1 && std::abs(delta) < std::abs(best_delta))) {
            best_distinct = distinct;
            best_delta = delta;
            best_sel = sel;
          }
        }
      }
      db.disc[h] = static_cast<uint16_t>((static_cast<uint32_t>(best_delta) & 0xFFu) | (best_sel << 8));
      for (size_t e = g0; e < g1; e++)
        keyed2.push_back({hg_disc_bucket(h, best_sel ? key_of(keyed[e].second, best_delta, best_sel) : 0u), keyed[e].second});
      g0 = g1;
    }
    std::stable_sort(keyed2.begin(), keyed2.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
    db.bucket_off2.assign((1u << HG_HASH_BITS) + 1, 0);
    db.windows2.clear();
    for (auto &kw : keyed2) {
      db.bucket_off2[kw.first + 1]++;
      db.windows2.push_back(kw.second);
    }
    for (size_t i = 1; i < db.bucket_off2.size(); i++) db.bucket_off2[i] += db.bucket_off2[i - 1];
    if (db.windows2.empty()) db.windows2.push_back(HgWindow{0, 0});
  }

  // Direct window table (hg_db.h HgWinBucket): every distinct window value once, with its (literal, offset) when there is only one.
  {
    std::vector<std::pair<uint32_t, uint32_t>> vf;  // (value, factor_off)
    for (auto &kw : keyed) vf.push_back({kw.second.value, kw.second.factor_off});
    std::sort(vf.begin(), vf.end());
    vf.erase(std::unique(vf.begin(), vf.end()), vf.end());
    size_t distinct = 0;
    for (size_t i = 0; i < vf.size(); i++) distinct += (i == 0 || vf[i].first != vf[i - 1].first) ? 1 : 0;
    uint32_t nbuckets = 16;
    while (static_cast<size_t>(nbuckets) * HG_WTAB_WAYS < 2 * distinct) nbuckets <<= 1;
    HgWinBucket empty;
    for (uint32_t k = 0; k < HG_WTAB_WAYS; k++) { empty.value[k] = 0; empty.factor_off[k] = HG_WTAB_EMPTY; }
    db.wtab.assign(nbuckets, empty);
    db.wtab_mask = nbuckets - 1;
    db.shared_windows = 0;
    for (size_t i = 0; i < vf.size();) {
      size_t j = i;
      while (j < vf.size() && vf[j].first == vf[i].first) j++;
      const uint32_t payload = j - i == 1 ? vf[i].second : HG_WTAB_SHARED;
      db.shared_windows += j - i == 1 ? 0 : 1;
      for (uint32_t b = hg_wtab_bucket(vf[i].first, db.wtab_mask);; b = (b + 1u) & db.wtab_mask) {
        bool placed_here = false;
        for (uint32_t k = 0; k < HG_WTAB_WAYS && !placed_here; k++)
          if (db.wtab[b].factor_off[k] == HG_WTAB_EMPTY) {
            db.wtab[b].value[k] = vf[i].first;
            db.wtab[b].factor_off[k] = payload;
            placed_here = true;
          }
        if (placed_here) break;
      }
      i = j;
    }
    db.wtab_first = (distinct > 0 && static_cast<size_t>(db.shared_windows) * 20 <= distinct) ? 1u : 0u;
  }

  // LDS filter over the distinct window values (single-probe slots below; two-slot cells in wide mode).
  std::vector<uint32_t> values;
  for (auto &kw : keyed) values.push_back(kw.second.value);
  std::sort(values.begin(), values.end());
  values.erase(std::unique(values.begin(), values.end()), values.end());
  // Single-probe slots.  Per table size and candidate weight vector the cost is the expected false-positive rate per
  // text dword (slot shared by windows whose fingerprints agree on b bits admits 2^-b of all dwords); the smallest
  // table that stays under the target wins: up to 16 KiB three stream workgroups fit on a CU, beyond that two or one.
  bool placed = false;
  {
    std::vector<uint32_t> cand_weights;
    for (uint32_t i = 0; i < HG_SLOT_WEIGHT_NCHOICES; i++) {
      cand_weights.push_back(HG_SLOT_WEIGHT_CHOICES[i][0] & wmask);  // (3-byte windows: the dword's top byte has weight zero)
      cand_weights.push_back(HG_SLOT_WEIGHT_CHOICES[i][1] & wmask);
    }
    auto slot_words = [&](uint32_t k, uint32_t weights, std::vector<uint32_t> &words) -> double {
      const uint32_t byte_mask = ((1u << k) - 1u) << 2;
      words.assign(size_t(1) << k, HG_FILTER_EMPTY);
      std::vector<uint8_t> used(size_t(1) << k, 0);
      for (uint32_t v : values) {
        const uint32_t sl = hg_slot(v, weights, byte_mask) >> 2, fp = hg_hash_window(v) & 0xFFFFu;
        uint32_t &w = words[sl];
        if (!used[sl]) {
          w = 0xFFFF0000u | fp;
          used[sl] = 1;
        } else {
          const uint32_t care = (w >> 16) & ~((w ^ fp) & 0xFFFFu);  // keep the bits on which all fingerprints agree
          w = (care << 16) | (w & care);
        }
      }
      double fp_rate = 0;
      for (size_t sl = 0; sl < words.size(); sl++)
        if (used[sl]) fp_rate += std::ldexp(1.0, -static_cast<int>(__builtin_popcount(words[sl] >> 16)));
      return fp_rate / static_cast<double>(words.size());
    };
    double best_rate = 1e9;
    uint32_t best_k = 0, best_w = 0;
    for (uint32_t k = HG_FILTER_MIN_LOG2; k <= HG_FILTER_MAX_LOG2 && !placed; k++) {
      if (values.size() > (size_t(1) << k)) continue;  // hopeless: more windows than slots
      double k_rate = 1e9;
      uint32_t k_w = 0;
      for (uint32_t weights : cand_weights) {
        std::vector<uint32_t> words;
        const double rate = slot_words(k, weights, words);
        if (rate < k_rate) { k_rate = rate; k_w = weights; }
      }
      if (k_rate < best_rate) { best_rate = k_rate; best_k = k; best_w = k_w; }
      if (k_rate <= (k <= 12 ? 5e-4 : 3e-4)) {
        best_rate = k_rate;
        best_k = k;
        best_w = k_w;
        placed = true;
      }
    }
    if (!placed && best_k && best_rate <= 2e-3 && values.size() * 100 <= (size_t(1) << HG_FILTER_MAX_LOG2) * 45) placed = true;  // dense, but still cheaper than wide mode
    if (placed) {
      const uint32_t k = best_k, wa = best_w;
      const uint32_t byte_mask = ((1u << k) - 1u) << 2;
      slot_words(k, wa, db.filter);
      db.ext.assign(size_t(1) << k, HgSlotInfo{});
      db.filter_log2 = k;
      db.weights_a = wa;
      db.weights_b = wa;  // unused outside wide mode
      // per slot: the neighbour-dword conditions (byte-wise agreement of all its windows)
      auto merge = [](uint32_t &val, uint32_t &mask, uint32_t v2, uint32_t m2) {
        uint32_t keep = 0;
        for (int b = 0; b < 4; b++) {
          uint32_t bm = 0xFFu << (8 * b);
          if ((mask & bm) && (m2 & bm) && ((val ^ v2) & bm) == 0) keep |= bm;
        }
        mask = keep;
        val &= keep;
      };
      for (size_t wi = 0; wi < keyed.size(); wi++) {
        const HgWindow &w = keyed[wi].second;
        const HgFactor &f = db.factors[w.factor_off >> 8];
        const int o = static_cast<int>(w.factor_off & 0xff);
        uint32_t pv = 0, pm = 0, nv = 0, nm = 0;
        for (int b = 0; b < 4; b++) {
          int jp = o - 4 + b, jn = o + static_cast<int>(wbytes) + b;
          if (jp >= 0) { pv |= static_cast<uint32_t>(f.lit[jp]) << (8 * b); pm |= 0xFFu << (8 * b); }
          if (jn < static_cast<int>(f.len)) { nv |= static_cast<uint32_t>(f.lit[jn]) << (8 * b); nm |= 0xFFu << (8 * b); }
        }
        pv = (pv | fold) & pm;
        nv = (nv | fold) & nm;
        if (!fold) {  // case-insensitive letters cannot be compared exactly without folding: drop them
          for (int b = 0; b < 4; b++) {
            int jp = o - 4 + b, jn = o + static_cast<int>(wbytes) + b;
            if (jp >= 0 && hg_factor_cmask(f, jp) != 0xFF) { pm &= ~(0xFFu << (8 * b)); pv &= pm; }
            if (jn < static_cast<int>(f.len) && hg_factor_cmask(f, jn) != 0xFF) { nm &= ~(0xFFu << (8 * b)); nv &= nm; }
          }
        }
        // the slot keeps up to two values exactly, each with the agreement of the conditions of ITS windows
        HgSlotInfo &info = db.ext[hg_slot(w.value, wa, byte_mask) >> 2];
        HgFilterExt *x = nullptr;
        bool fresh = false;
        for (uint32_t i = 0; i < info.nvalues; i++)
          if (info.value[i] == w.value) x = &info.cond[i];
        if (!x && info.nvalues < 2) {
          info.value[info.nvalues] = w.value;
          x = &info.cond[info.nvalues++];
          fresh = true;
        }
        if (!x) {
          x = &info.rest;
          fresh = !info.many;
          info.many = 1;
        }
        if (fresh) {
          *x = HgFilterExt{pv, pm, nv, nm};
        } else {
          merge(x->pv, x->pm, pv, pm);
          merge(x->nv, x->nm, nv, nm);
        }
      }
    }
  }
  db.filter_wide = 0;
  if (!placed && db.dense) {
    if (err) *err = "too many literal windows for byte-aligned probing";
    return -5;  // the caller falls back to dword-aligned windows
  }
  if (!placed) {
    // Wide mode for large pattern sets: every 4-byte slot holds TWO 16-bit fingerprints (cells), each window may sit
    // in either cell of either of its two slots (bucketed cuckoo, usable up to ~85 % of the cells), and the neighbour
    // conditions are not used (a 16-bit fingerprint hit goes straight to the verify pass).
    for (uint32_t attempt = 0; attempt < 3 * HG_SLOT_WEIGHT_NCHOICES && !placed; attempt++) {
      const uint32_t k = 13 + attempt / HG_SLOT_WEIGHT_NCHOICES;  // 32 / 64 / 128 KiB of LDS
      const uint32_t wa = HG_SLOT_WEIGHT_CHOICES[attempt % HG_SLOT_WEIGHT_NCHOICES][0], wb = HG_SLOT_WEIGHT_CHOICES[attempt % HG_SLOT_WEIGHT_NCHOICES][1];
      const size_t cells = size_t(2) << k;
      if (values.size() * 100 > cells * 85) continue;
      const uint32_t byte_mask = ((1u << k) - 1u) << 2;
      struct WEntry { uint32_t sa, sb; uint16_t fp; };
      std::vector<WEntry> entries;
      {
        std::vector<std::array<uint32_t, 3>> keys;
        for (uint32_t v : values) {
          const uint32_t a = hg_dot4(v, wa), b = hg_dot4(v, wb);
          const uint32_t sa = hg_slot_wide(a, b, byte_mask) >> 2, sb = hg_slot_wide(b, a, byte_mask) >> 2;
          keys.push_back({std::min(sa, sb), std::max(sa, sb), hg_hash_window(v) & 0xFFFFu});
        }
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        for (auto &key : keys) entries.push_back({key[0], key[1], static_cast<uint16_t>(key[2])});
      }
      std::vector<int32_t> owner(cells, -1);  // cell = slot * 2 + {0, 1}
      uint64_t rng = 0x9E3779B97F4A7C15ull;
      placed = true;
      for (size_t e = 0; e < entries.size() && placed; e++) {
        int32_t cur = static_cast<int32_t>(e);
        bool ok = false;
        for (int kick = 0; kick < 8000 && !ok; kick++) {
          const uint32_t cand[4] = {entries[cur].sa * 2, entries[cur].sa * 2 + 1, entries[cur].sb * 2, entries[cur].sb * 2 + 1};
          for (uint32_t cell : cand)
            if (owner[cell] < 0) { owner[cell] = cur; ok = true; break; }
          if (ok) break;
          rng = rng * 6364136223846793005ull + 1442695040888963407ull;
          std::swap(cur, owner[cand[(rng >> 33) & 3]]);
        }
        if (!ok) placed = false;
      }
      if (!placed) continue;
      db.filter.assign(size_t(1) << k, HG_FILTER_EMPTY);
      for (size_t cell = 0; cell < cells; cell++) {
        if (owner[cell] < 0) continue;
        uint32_t &word = db.filter[cell >> 1];
        const uint32_t fp = entries[owner[cell]].fp;
        word = (cell & 1) ? ((word & 0x0000FFFFu) | (fp << 16)) : ((word & 0xFFFF0000u) | fp);
      }
      db.ext.assign(1, HgSlotInfo{});
      db.filter_log2 = k;
      db.filter_wide = 1;
      db.weights_a = wa;
      db.weights_b = wb;
    }
  }
  if (!placed) {
    if (err) *err = "too many distinct literal windows for the LDS filter (more than ~55 000)";
    return -4;
  }
  return 0;
}

// Packs single-word always-on expressions into shared state words (HgSlowGroup): first fit by node count, the expressions
// with boundary conditions first (their bins run the routine with conditions, and whatever context-free expression still
// fits rides along), then the context-free ones (bins of their own once the others are full).
// A/B knobs of the compiler (older code paths, for experiments and one test), read from the environment once per compile.
struct CompileKnobs {
  bool no_slow_groups = std::getenv("HG_NO_SLOW_GROUPS") != nullptr;
  bool no_ctx_groups = std::getenv("HG_NO_CTX_GROUPS") != nullptr;
  bool no_confirm_window = std::getenv("HG_NO_CONFIRM_WINDOW") != nullptr;
  bool no_byte_windows = std::getenv("HG_NO_BYTE_WINDOWS") != nullptr;
  bool no_case_expand = std::getenv("HG_NO_CASE_EXPAND") != nullptr;
};

void build_slow_groups(HgDb &db, const CompileKnobs &knobs) {
  db.groups.clear();
  db.nslow_grouped = 0;
  if (knobs.no_slow_groups) return;
  const bool mixed = !knobs.no_ctx_groups;
  struct Bin { std::vector<uint32_t> members; uint32_t nodes = 0; bool ctx = false; };
  std::vector<Bin> bins;
  for (int pass = 0; pass < 2; pass++) {  // 0: expressions with conditions, 1: context-free ones
    for (uint32_t j = 0; j < db.nslow_fast; j++) {
      const HgPattern &p = db.patterns[db.slow[j]];
      if (p.nw != 1 || p.nnodes == 0 || (p.simple != 0) != (pass == 1)) continue;
      if (pass == 0 && !mixed) continue;
      Bin *home = nullptr;
      for (Bin &b : bins)
        if (b.nodes + p.nnodes <= 32 && b.members.size() < HG_GROUP_MAX_MEMBERS) { home = &b; break; }
      if (!home) { bins.emplace_back(); home = &bins.back(); home->ctx = pass == 0; }
      home->members.push_back(db.slow[j]);
      home->nodes += p.nnodes;
    }
  }
  std::vector<uint32_t> grouped;
  for (const Bin &b : bins) {
    if (b.members.size() < 2) continue;
    HgSlowGroup g{};
    db.pool.resize((db.pool.size() + 3) & ~static_cast<size_t>(3), 0);  // (the wave stages reach[] with 16-byte loads)
    g.reach_off = static_cast<uint32_t>(db.pool.size());
    db.pool.resize(db.pool.size() + 256, 0);
    g.follow_off = static_cast<uint32_t>(db.pool.size());
    db.pool.resize(db.pool.size() + b.nodes, 0);
    if (b.ctx) {
      g.ctx_off = static_cast<uint32_t>(db.pool.size());
      db.pool.resize(db.pool.size() + 36, 0);
    }
    g.nnodes = b.nodes;
    g.nmembers = static_cast<uint32_t>(b.members.size());
    bool unbounded = false;
    uint32_t shift = 0;
    for (uint32_t m = 0; m < g.nmembers; m++) {
      const HgPattern &p = db.patterns[b.members[m]];
      const uint32_t all = (p.nnodes == 32 ? 0xFFFFFFFFu : ((1u << p.nnodes) - 1u)) << shift;
      g.member[m] = b.members[m];
      g.acc[m] = b.ctx ? all : p.acc_all << shift;
      g.acc_all |= g.acc[m];
      g.init_word |= db.pool[p.init_off] << shift;
      if (p.single) g.single_mask |= 1u << m;
      if (p.max_len == 0) unbounded = true;
      g.max_len = std::max(g.max_len, p.max_len);
      for (uint32_t c = 0; c < 256; c++) db.pool[g.reach_off + c] |= db.pool[p.reach_off + c] << shift;
      for (uint32_t v = 0; v < p.nnodes; v++) db.pool[g.follow_off + shift + v] = db.pool[p.follow_off + v] << shift;
      if (b.ctx) {
        for (uint32_t i = 0; i < 16; i++) db.pool[g.ctx_off + i] |= db.pool[p.amask_off + i] << shift;
        for (uint32_t i = 0; i < 20; i++) db.pool[g.ctx_off + 16 + i] |= db.pool[p.acc_off + i] << shift;
      }
      shift += p.nnodes;
      grouped.push_back(b.members[m]);
    }
    if (unbounded) g.max_len = 0;
    db.groups.push_back(g);
  }
  if (grouped.empty()) return;
  // members first (group by group), then the other fast entries, then the rest: the order inside each class is kept
  std::vector<uint32_t> rest_fast, rest_slow;
  for (uint32_t j = 0; j < db.slow.size(); j++) {
    const uint32_t pi = db.slow[j];
    if (std::find(grouped.begin(), grouped.end(), pi) != grouped.end()) continue;
    (j < db.nslow_fast ? rest_fast : rest_slow).push_back(pi);
  }
  db.slow = grouped;
  db.slow.insert(db.slow.end(), rest_fast.begin(), rest_fast.end());
  db.slow.insert(db.slow.end(), rest_slow.begin(), rest_slow.end());
  db.nslow_grouped = static_cast<uint32_t>(grouped.size());
}

}  // namespace

int hgc_compile(const char *const *exprs, const unsigned *flags, const unsigned *ids, unsigned n, HgDb **out,
               std::string *err, int *bad_index) {
  if (bad_index) *bad_index = -1;
  if (out) *out = nullptr;
  if (!exprs || !out || n == 0) {
    if (err) *err = "invalid arguments: at least one expression is required";
    return -1;
  }
  if (n > HG_MAX_PATTERNS) {
    if (err) *err = "too many expressions (limit 16777216)";
    return -4;
  }
  const CompileKnobs knobs;
  auto db = std::make_unique<HgDb>();
  struct Pending { std::vector<Lit> lits; bool literal_only = false; };
  std::vector<Pending> covers(n);
  unsigned cur = 0;
  try {
    for (cur = 0; cur < n; cur++) {
      uint32_t f = flags ? flags[cur] : 0;
      if (f & ~HG_FLAGS_SUPPORTED) throw CompileError("unsupported flag bits");
      if (!exprs[cur] || !exprs[cur][0]) throw CompileError("empty expression");
      std::string text(exprs[cur]);
      db->exprs.push_back(text);
      Parser parser(text, f);
      NodeP root = parser.parse();
      check_embedded_anchors(*root, false, false);

      if (program_size(*root) > HG_HUGE_MAX_PROGRAM) throw CompileError("pattern too large");
      Glushkov g;
      Frag top = g.build(*root);
      if (top.nullable) throw CompileError("expression can match the empty string (HS_FLAG_ALLOWEMPTY is not supported)");
      if (g.pos_class.empty()) throw CompileError("expression matches nothing");

      // one edge per (p, q): the ways of getting from p to q (through different assertions) merge into one condition
      std::sort(g.edges.begin(), g.edges.end(), [](const Edge &x, const Edge &y) { return x.p != y.p ? x.p < y.p : x.q < y.q; });
      {
        size_t out_n = 0;
        for (size_t i = 0; i < g.edges.size(); i++) {
          if (out_n && g.edges[out_n - 1].p == g.edges[i].p && g.edges[out_n - 1].q == g.edges[i].q) g.edges[out_n - 1].tt |= g.edges[i].tt;
          else g.edges[out_n++] = g.edges[i];
        }
        g.edges.resize(out_n);
      }
      // nodes = distinct (position, entry condition), numbered in (position, condition) order: linear expressions get
      // follow = next bit, and the nodes of one position are consecutive
      std::vector<std::pair<uint32_t, uint32_t>> nodes;
      nodes.reserve(top.first.size() + g.edges.size());
      for (auto &c : top.first) nodes.push_back({c.pos, c.tt});
      for (auto &e : g.edges) nodes.push_back({e.q, e.tt});
      std::sort(nodes.begin(), nodes.end());
      nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
      if (nodes.size() > HG_HUGE_MAX_NODES) throw CompileError("pattern too large");
      auto intern = [&](uint32_t pos, uint32_t tt) {
        return static_cast<uint32_t>(std::lower_bound(nodes.begin(), nodes.end(), std::make_pair(pos, tt)) - nodes.begin());
      };
      const uint32_t npos = static_cast<uint32_t>(g.pos_class.size());
      std::vector<uint32_t> node_lo(npos + 1, 0);  // nodes of position q: [node_lo[q], node_lo[q + 1])
      for (auto &nd : nodes) node_lo[nd.first + 1]++;
      for (uint32_t q = 0; q < npos; q++) node_lo[q + 1] += node_lo[q];

      // (an expression whose every entry condition is contradictory, e.g. \b\Bc, has no nodes: it keeps one all-zero state
      // word so that every routine sees well-formed tables and simply never matches)
      uint32_t nn = static_cast<uint32_t>(nodes.size()), nw = nn ? (nn + 31) / 32 : 1;
      HgPattern p{};
      p.id = ids ? ids[cur] : 0;
      p.flags = f;
      p.nnodes = nn;
      p.nw = nw;
      p.single = (f & HG_FLAG_SINGLEMATCH) ? 1 : 0;
      auto alloc = [&](size_t words) {
        if (db->pool.size() + words > 0xFFFF0000ull) throw CompileError("pattern set too large");
        uint32_t off = static_cast<uint32_t>(db->pool.size());
        db->pool.resize(db->pool.size() + words, 0);
        return off;
      };
      std::vector<uint32_t> last_tt(npos, 0);
      for (auto &l : top.last) last_tt[l.pos] |= l.tt;
      auto setbit = [&](uint32_t base, uint32_t node) { db->pool[base + node / 32] |= 1u << (node % 32); };
      const bool huge = nw > HG_MAX_W;
      if (!huge) {
        p.reach_off = alloc(256 * nw);
        p.follow_off = alloc(static_cast<size_t>(nn) * nw);
        p.init_off = alloc(nw);
        p.amask_off = alloc(16 * nw);
        p.acc_off = alloc(20 * nw);
        for (uint32_t v = 0; v < nn; v++) {
          uint32_t pos = nodes[v].first, tt = nodes[v].second;
          for (int b = 0; b < 256; b++)
            if (g.pos_class[pos][b]) setbit(p.reach_off + b * nw, v);
          for (uint32_t pc = 0; pc < 4; pc++) {
            for (uint32_t cc = 0; cc < 4; cc++)
              if (tt >> (pc * 5 + cc) & 1) setbit(p.amask_off + (pc * 4 + cc) * nw, v);
            for (uint32_t nc = 0; nc < 5; nc++)
              if (last_tt[pos] >> (pc * 5 + nc) & 1) setbit(p.acc_off + (pc * 5 + nc) * nw, v);
          }
        }
        for (auto &c : top.first) setbit(p.init_off, intern(c.pos, c.tt));
        for (auto &e : g.edges) {
          uint32_t to = intern(e.q, e.tt);
          for (uint32_t v = node_lo[e.p]; v < node_lo[e.p + 1]; v++) setbit(p.follow_off + v * nw, to);
        }
      } else {
        // ---- huge automaton: sparse tables (HgHugeHeader, hg_db.h)
        db->nhuge++;
        db->huge_max_nw = std::max(db->huge_max_nw, nw);
        // byte classes: bytes that belong to the same position classes are one class
        std::vector<ByteSet> distinct;
        {
          std::map<std::string, uint32_t> seen;
          for (auto &cs : g.pos_class) {
            std::string key = cs.to_string();
            if (seen.emplace(key, 0).second) distinct.push_back(cs);
          }
        }
        uint32_t cls_of[256], ncls = 0;
        uint32_t rep[256];  // a representative byte of each class
        {
          std::map<std::vector<bool>, uint32_t> sig_cls;
          for (int b = 0; b < 256; b++) {
            std::vector<bool> sig(distinct.size());
            for (size_t d = 0; d < distinct.size(); d++) sig[d] = distinct[d][b];
            auto it = sig_cls.find(sig);
            if (it == sig_cls.end()) {
              rep[ncls] = static_cast<uint32_t>(b);
              it = sig_cls.emplace(std::move(sig), ncls++).first;
            }
            cls_of[b] = it->second;
          }
        }
        bool ctxfree = true;
        for (uint32_t v = 0; v < nn; v++) ctxfree = ctxfree && nodes[v].second == HG_TT_ALL;
        for (uint32_t q = 0; q < npos; q++) ctxfree = ctxfree && (last_tt[q] == 0 || last_tt[q] == HG_TT_ALL);
        const uint32_t hdr_off = alloc(sizeof(HgHugeHeader) / 4);
        const uint32_t cls_off = alloc(64);
        for (int b = 0; b < 256; b++) db->pool[cls_off + b / 4] |= cls_of[b] << ((b & 3) * 8);
        p.reach_off = alloc(static_cast<size_t>(ncls) * nw);
        p.follow_off = hdr_off;
        p.init_off = alloc(nw);
        const uint32_t smask_off = alloc(nw), xsrc_off = alloc(nw), xrank_off = alloc(nw);
        p.amask_off = ctxfree ? p.init_off : alloc(16 * static_cast<size_t>(nw));  // (never read when ctxfree)
        p.acc_off = alloc((ctxfree ? 1 : 20) * static_cast<size_t>(nw));
        for (uint32_t v = 0; v < nn; v++) {
          const uint32_t pos = nodes[v].first, tt = nodes[v].second;
          for (uint32_t c = 0; c < ncls; c++)
            if (g.pos_class[pos][rep[c]]) setbit(p.reach_off + c * nw, v);
          if (ctxfree) {
            if (last_tt[pos]) setbit(p.acc_off, v);
            continue;
          }
          for (uint32_t pc = 0; pc < 4; pc++) {
            for (uint32_t cc = 0; cc < 4; cc++)
              if (tt >> (pc * 5 + cc) & 1) setbit(p.amask_off + (pc * 4 + cc) * nw, v);
            for (uint32_t nc = 0; nc < 5; nc++)
              if (last_tt[pos] >> (pc * 5 + nc) & 1) setbit(p.acc_off + (pc * 5 + nc) * nw, v);
          }
        }
        for (auto &c : top.first) setbit(p.init_off, intern(c.pos, c.tt));
        uint32_t init_hi = 0;
        for (uint32_t w = 0; w < nw; w++)
          if (db->pool[p.init_off + w]) init_hi = w + 1;
        // follow: per source node its sorted targets; node -> node + 1 is a bit of smask, the rest become ranges
        std::vector<uint32_t> xlist{0}, xt;
        std::vector<uint32_t> targets;
        size_t ei = 0;
        for (uint32_t q = 0; q < npos; q++) {
          targets.clear();
          for (; ei < g.edges.size() && g.edges[ei].p == q; ei++) targets.push_back(intern(g.edges[ei].q, g.edges[ei].tt));
          if (targets.empty()) continue;
          std::sort(targets.begin(), targets.end());
          targets.erase(std::unique(targets.begin(), targets.end()), targets.end());
          for (uint32_t v = node_lo[q]; v < node_lo[q + 1]; v++) {  // every node of the position has the position's targets
            bool any = false;
            for (size_t t = 0; t < targets.size();) {
              if (targets[t] == v + 1) { setbit(smask_off, v); t++; continue; }
              size_t u = t;
              while (u + 1 < targets.size() && targets[u + 1] == targets[u] + 1 && targets[u + 1] != v + 1) u++;
              xt.push_back(targets[t]);
              xt.push_back(targets[u]);
              any = true;
              t = u + 1;
            }
            if (any) {
              setbit(xsrc_off, v);
              xlist.push_back(static_cast<uint32_t>(xt.size() / 2));
            }
          }
        }
        for (uint32_t w = 0, run = 0; w < nw; w++) {
          db->pool[xrank_off + w] = run;
          run += static_cast<uint32_t>(__builtin_popcount(db->pool[xsrc_off + w]));
        }
        const uint32_t xlist_off = alloc(xlist.size()), xt_off = alloc(std::max<size_t>(xt.size(), 2));
        std::copy(xlist.begin(), xlist.end(), db->pool.begin() + xlist_off);
        std::copy(xt.begin(), xt.end(), db->pool.begin() + xt_off);
        HgHugeHeader h{};
        h.cls_off = cls_off;
        h.ncls = ncls;
        h.smask_off = smask_off;
        h.xsrc_off = xsrc_off;
        h.xrank_off = xrank_off;
        h.xlist_off = xlist_off;
        h.xt_off = xt_off;
        h.ctxfree = ctxfree ? 1u : 0u;
        h.init_hi = init_hi;
        h.nsources = static_cast<uint32_t>(xlist.size() - 1);
        h.nranges = static_cast<uint32_t>(xt.size() / 2);
        std::memcpy(&db->pool[hdr_off], &h, sizeof h);
        const uint64_t stage = 64ull + static_cast<uint64_t>(nw) * (ncls + 4u + (ctxfree ? 1u : 36u));  // (hg_huge.hip huge_stage_words)
        if (stage <= HG_HUGE_STAGE_MAX) db->huge_stage_words = std::max<uint32_t>(db->huge_stage_words, static_cast<uint32_t>(stage));
      }
      if (!huge) db->max_nw = std::max(db->max_nw, nw);
      db->max_id = std::max(db->max_id, p.id);
      if (nw == 1) {  // context-free single-word automaton: the confirm kernel's fast path
        bool simple = true;
        const uint32_t full = nn == 32 ? 0xFFFFFFFFu : ((1u << nn) - 1u);
        for (uint32_t i = 0; i < 16; i++) simple = simple && db->pool[p.amask_off + i] == full;
        for (uint32_t i = 1; i < 20; i++) simple = simple && db->pool[p.acc_off + i] == db->pool[p.acc_off];
        p.simple = simple ? 1 : 0;
        p.acc_all = db->pool[p.acc_off];
        p.init_word = db->pool[p.init_off];
      }

      // required literals
      Info info = analyze(*root);
      LitSet cover;
      long lead = info.lead;  // (a literal cut down to HG_FACTOR_MAX bytes begins that much later)
      if (info.has_cover)
        for (auto &l : info.cover) {
          size_t cut = 0;
          cover.push_back(clip(l, &cut));
          if (cut) lead = add_len(lead, static_cast<long>(cut));
        }
      dedupe(cover);
      covers[cur].lits = cover;
      p.lit_lead = (info.has_cover && lead >= 0 && !knobs.no_confirm_window) ? static_cast<uint32_t>(lead) : 0xFFFFFFFFu;
      // literal-only: the expression's language is exactly one literal that fits the factor record, has no NUL
      // or inner newline, and no assertions -> a verified factor occurrence is a match
      if (info.exact && info.set.size() == 1 && cover.size() == 1 && info.set[0].bytes.size() <= HG_FACTOR_MAX &&
          cover[0] == info.set[0] && !has_assert(*root)) {
        const std::string &lb = info.set[0].bytes;
        bool clean = true;
        for (size_t j = 0; j < lb.size(); j++)
          if (lb[j] == 0 || (lb[j] == '\n' && j + 1 < lb.size())) clean = false;
        covers[cur].literal_only = clean;
      }
      {
        const long ml = max_match_len(*root);
        p.max_len = ml > 0 ? static_cast<uint32_t>(ml) : 0;
      }
      db->patterns.push_back(p);
    }
  } catch (const CompileError &e) {
    if (err) *err = e.what();
    if (bad_index) *bad_index = static_cast<int>(cur);
    return -4;
  } catch (const std::bad_alloc &) {
    if (err) *err = "out of memory";
    if (bad_index) *bad_index = static_cast<int>(cur);
    return -2;
  }

  // Tiers.  A pattern whose required literals all have at least `min_factor` bytes is found through the window prefilter
  // (tier 0), the others run on every line (tier 1).  Dword-aligned windows need HG_FAST_MIN_FACTOR bytes (a window on
  // every residue mod 4).  When that leaves patterns with shorter literals behind, the stream pass probes a window at
  // every BYTE offset instead (db.dense: four times the probes, a third of the streaming rate — still several times the
  // always-on tier), which takes literals down to HG_DENSE_MIN_FACTOR bytes.
  auto assign = [&](uint32_t min_factor, uint32_t dense, uint32_t window_bytes) -> int {
    db->dense = dense;
    db->window_bytes = window_bytes;
    db->window_mask = window_bytes == 4 ? 0xFFFFFFFFu : 0x00FFFFFFu;
    db->weights_c = HG_HASH_WEIGHTS & db->window_mask;
    db->slow.clear();
    db->factors.clear();
    db->fold_mask = 0;
    size_t nlits = 0, ncaseless = 0;  // required literals of the anchored expressions / those with case-insensitive letters
    for (uint32_t m = 0; m < HG_CONFIRM_MODES; m++) db->n_confirm_mode[m] = 0;
    for (unsigned i = 0; i < n; i++) {
      HgPattern &p = db->patterns[i];
      const LitSet &cover = covers[i].lits;
      const bool fast = !cover.empty() && min_len(cover) >= min_factor;
      p.tier = fast ? 0 : 1;
      p.literal_only = (fast && covers[i].literal_only) ? 1 : 0;
      if (fast) {
        db->n_confirm_mode[hg_confirm_mode(p)]++;
        for (auto &l : cover) {
          nlits++;
          bool caseless = false;
          for (unsigned char m : l.cmask) caseless = caseless || m != 0xFF;
          ncaseless += caseless ? 1 : 0;
        }
      } else {
        db->slow.push_back(i);
      }
    }
    // Case-insensitive literals.  The general way: the stream pass folds every text dword (| 0x20202020) before it hashes it, and the
    // windows are stored folded.  Where only a few literals of a set of dword-aligned windows are case-insensitive, their windows are
    // stored in every case variant instead (at most 16 per window) and NOTHING is folded: the hot loop saves an instruction per
    // dword, and folded look-alikes ('@' / '`', '[' / '{', upper-case text) no longer pass the filter.
    if (ncaseless) {
      const bool expand = !dense && !knobs.no_case_expand && ncaseless <= 64;  // (at most 64 x 4 windows x 16 variants more)
      (void)nlits;
      db->fold_mask = expand ? 0u : 0x20202020u;
    }
    // always-on patterns of at most two state words go first: the segment-parallel kernel takes those
    auto two_words = [&](uint32_t pi) { return db->patterns[pi].nw <= 2; };  // bounded or not: an unbounded pattern's lead-in is the start of its line
    auto not_huge = [&](uint32_t pi) { return db->patterns[pi].nw <= HG_MAX_W; };  // huge automata go last: their own kernel
    std::stable_partition(db->slow.begin(), db->slow.end(), not_huge);
    std::stable_partition(db->slow.begin(), db->slow.end(), two_words);
    db->nslow_fast = static_cast<uint32_t>(std::count_if(db->slow.begin(), db->slow.end(), two_words));
    db->nslow_huge = static_cast<uint32_t>(db->slow.size() - std::count_if(db->slow.begin(), db->slow.end(), not_huge));
    build_slow_groups(*db, knobs);
    // factors (needs the final fold mask); windows and filter tables are built from them
    uint32_t rank_in_mode[HG_CONFIRM_MODES] = {};
    for (unsigned i = 0; i < n; i++) {
      if (db->patterns[i].tier != 0) continue;
      const uint32_t mode = hg_confirm_mode(db->patterns[i]), rank = rank_in_mode[mode]++;
      for (auto &l : covers[i].lits) {
        HgFactor fct{};
        fct.pattern = i;
        fct.len = static_cast<uint32_t>(l.bytes.size());
        fct.mode = mode;
        fct.mode_rank = rank;
        std::memcpy(fct.lit, l.bytes.data(), fct.len);
        for (uint32_t b = 0; b < fct.len; b++)
          if (static_cast<unsigned char>(l.cmask[b]) != 0xFF) fct.casebits |= 1u << b;
        fct.id = db->patterns[i].id;
        db->factors.push_back(fct);
      }
    }
    db->nreal_factors = static_cast<uint32_t>(db->factors.size());
    if (db->factors.empty()) db->factors.push_back(HgFactor{});  // keep device arrays non-empty
    return build_filter(*db, nullptr, err);
  };
  size_t shortest = SIZE_MAX;  // shortest required literal among the patterns dword-aligned windows leave behind
  for (unsigned i = 0; i < n; i++) {
    const size_t m = covers[i].lits.empty() ? 0 : min_len(covers[i].lits);
    if (m >= HG_DENSE_MIN_FACTOR && m < HG_FAST_MIN_FACTOR) shortest = std::min(shortest, m);
  }
  int rc = -5;
  if (shortest != SIZE_MAX && !knobs.no_byte_windows) {
    // every literal of the set has at least 5 bytes: a window on both residues mod 2 fits, half the probes
    size_t set_shortest = SIZE_MAX;
    for (unsigned i = 0; i < n; i++)
      if (!covers[i].lits.empty() && min_len(covers[i].lits) >= shortest) set_shortest = std::min(set_shortest, min_len(covers[i].lits));
    const uint32_t step = set_shortest >= HG_WINDOW_BYTES + 1 ? 2u : 1u;
    rc = assign(static_cast<uint32_t>(shortest), step, HG_WINDOW_BYTES);
    if (rc == -5 && step == 2) rc = assign(static_cast<uint32_t>(shortest), 1, HG_WINDOW_BYTES);  // (one window per literal instead of two)
    // too many 3-byte literals to enumerate the byte after each: every window shrinks to 3 bytes instead
    if (rc == -5 && shortest < HG_WINDOW_BYTES) rc = assign(static_cast<uint32_t>(shortest), 1, HG_WINDOW_BYTES - 1);
    if (rc == -5 && shortest < HG_WINDOW_BYTES) rc = assign(HG_WINDOW_BYTES, 1, HG_WINDOW_BYTES);  // without the short literals
  }
  if (rc == -5) rc = assign(HG_FAST_MIN_FACTOR, 0, HG_WINDOW_BYTES);
  if (rc != 0) {
    if (bad_index) *bad_index = -1;
    return -4;
  }
  *out = db.release();
  return 0;
}

int hgc_tune(const HgDb *db, const uint8_t *sample, size_t nbytes, HgDb **out, std::string *err) {
  if (out) *out = nullptr;
  if (!db || !out || (!sample && nbytes)) return -1;
  SampleStats st;
  const uint32_t fold = db->fold_mask;
  for (size_t p = 0; p + 4 <= nbytes; p += db->dense ? db->dense : 4) {
    uint32_t w, nx = 0;
    std::memcpy(&w, sample + p, 4);
    if (p + 8 <= nbytes) std::memcpy(&nx, sample + p + 4, 4);
    w = (w | fold) & db->window_mask;
    nx |= fold;
    st.c4[w]++;
    st.c6[static_cast<uint64_t>(w) | (static_cast<uint64_t>(nx & 0xFFFFu) << 32)]++;
  }
  st.dwords = nbytes / 4;
  std::unique_ptr<HgDb> copy;
  try {
    copy = std::make_unique<HgDb>(*db);
  } catch (const std::bad_alloc &) {
    if (err) *err = "out of memory";
    return -2;
  }
  int rc = build_filter(*copy, &st, err);  // a failure leaves the caller's database as it was
  if (rc != 0) return rc;
  copy->tuned = true;
  *out = copy.release();
  return 0;
}

void hgc_free(HgDb *db) { delete db; }
