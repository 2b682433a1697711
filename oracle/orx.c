/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see orx.h).  Plain C restatement of the regex semantics
 * behind hs_compile_multi / block-mode hs_scan as used at
 * /root/reference/hypergrep/lib/c/hyperscanner.c:136,217 (Intel Hyperscan 5.4.2, third-party).
 */
#include "orx.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ byte classes */
typedef struct {
    uint32_t w[8];
} cls_t;

static void cls_set(cls_t *c, unsigned b) { c->w[b >> 5] |= 1u << (b & 31); }
static int cls_has(const cls_t *c, unsigned b) { return (c->w[b >> 5] >> (b & 31)) & 1u; }
static void cls_range(cls_t *c, unsigned lo, unsigned hi) {
    for (unsigned b = lo; b <= hi; b++) cls_set(c, b);
}
static void cls_or(cls_t *c, const cls_t *o) {
    for (int i = 0; i < 8; i++) c->w[i] |= o->w[i];
}
static void cls_not(cls_t *c) {
    for (int i = 0; i < 8; i++) c->w[i] = ~c->w[i];
}
static int cls_empty(const cls_t *c) {
    for (int i = 0; i < 8; i++)
        if (c->w[i]) return 0;
    return 1;
}
static void cls_caseless(cls_t *c) {
    for (unsigned b = 'a'; b <= 'z'; b++) {
        if (cls_has(c, b)) cls_set(c, b - 32);
        if (cls_has(c, b - 32)) cls_set(c, b);
    }
}
static int is_word(unsigned b) {
    return (b >= '0' && b <= '9') || (b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z') || b == '_';
}
static void cls_digit(cls_t *c) { cls_range(c, '0', '9'); }
static void cls_word(cls_t *c) {
    cls_range(c, '0', '9');
    cls_range(c, 'A', 'Z');
    cls_range(c, 'a', 'z');
    cls_set(c, '_');
}
static void cls_space(cls_t *c) { /* PCRE >= 8.34: includes VT */
    cls_set(c, ' ');
    cls_range(c, 9, 13);
}
static void cls_hspace(cls_t *c) {
    cls_set(c, 9);
    cls_set(c, ' ');
    cls_set(c, 0xA0);
}
static void cls_vspace(cls_t *c) {
    cls_range(c, 10, 13);
    cls_set(c, 0x85);
}

/* ------------------------------------------------------------------ AST */
enum { N_EMPTY, N_CLASS, N_CAT, N_ALT, N_REP, N_ASSERT };
enum { A_BOL_ML, A_BOL, A_EOL_ML, A_EOL, A_EOD, A_WB, A_NWB };

typedef struct node {
    int type;
    cls_t cls;
    int akind;
    int min, max; /* max < 0: unbounded */
    struct node **kids;
    int nkids;
} node;

#define F_I 1u
#define F_S 2u
#define F_M 4u
#define F_X 16u /* internal: extended / free-spacing */

typedef struct {
    const unsigned char *p, *end;
    char *err;
    size_t errlen;
    int failed;
    int depth;
} parser;

static void fail(parser *ps, const char *fmt, ...) {
    if (ps->failed) return;
    ps->failed = 1;
    if (ps->err && ps->errlen) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ps->err, ps->errlen, fmt, ap);
        va_end(ap);
    }
}

static node *mk(int type) {
    node *n = (node *)calloc(1, sizeof(node));
    n->type = type;
    return n;
}
static void add_kid(node *n, node *k) {
    n->kids = (node **)realloc(n->kids, sizeof(node *) * (size_t)(n->nkids + 1));
    n->kids[n->nkids++] = k;
}
static void free_node(node *n) {
    if (!n) return;
    for (int i = 0; i < n->nkids; i++) free_node(n->kids[i]);
    free(n->kids);
    free(n);
}
static int hexval(int c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

/* Parse an escape that denotes a single byte (after the backslash, *ps->p is the escape char).
 * Returns byte value or -1 if not a single-byte escape (p left unchanged in that case). */
static int parse_byte_escape(parser *ps, int in_class) {
    const unsigned char *q = ps->p;
    if (q >= ps->end) return -1;
    int c = *q;
    switch (c) {
    case 'n': ps->p++; return '\n';
    case 'r': ps->p++; return '\r';
    case 't': ps->p++; return '\t';
    case 'f': ps->p++; return '\f';
    case 'a': ps->p++; return 7;
    case 'e': ps->p++; return 27;
    case 'b':
        if (in_class) { ps->p++; return 8; }
        return -1;
    case 'c':
        if (q + 1 < ps->end) {
            int x = q[1];
            if (x >= 'a' && x <= 'z') x -= 32;
            ps->p += 2;
            return x ^ 0x40;
        }
        fail(ps, "\\c at end of pattern");
        return -1;
    case 'x': {
        q++;
        if (q < ps->end && *q == '{') {
            q++;
            unsigned v = 0;
            int nd = 0;
            while (q < ps->end && hexval(*q) >= 0) {
                v = v * 16 + (unsigned)hexval(*q);
                if (v > 0xFF) { fail(ps, "\\x{} value too large for byte mode"); return -1; }
                q++;
                nd++;
            }
            if (q >= ps->end || *q != '}' || nd == 0) { fail(ps, "malformed \\x{}"); return -1; }
            ps->p = q + 1;
            return (int)v;
        }
        unsigned v = 0;
        int nd = 0;
        while (nd < 2 && q < ps->end && hexval(*q) >= 0) {
            v = v * 16 + (unsigned)hexval(*q);
            q++;
            nd++;
        }
        ps->p = q;
        return (int)v;
    }
    case '0': case '1': case '2': case '3': case '4': case '5': case '6': case '7': {
        /* \0 and \0dd are octal; \ddd (three octal digits) is octal; otherwise \1-\9 are backrefs. */
        int is_octal = 0;
        if (c == '0' || in_class) is_octal = 1;
        else if (q + 2 < ps->end && q[1] >= '0' && q[1] <= '7' && q[2] >= '0' && q[2] <= '7') is_octal = 1;
        if (!is_octal) return -1;
        unsigned v = 0;
        int nd = 0;
        while (nd < 3 && q < ps->end && *q >= '0' && *q <= '7') {
            v = v * 8 + (unsigned)(*q - '0');
            q++;
            nd++;
        }
        if (v > 0xFF) { fail(ps, "octal value too large"); return -1; }
        ps->p = q;
        return (int)v;
    }
    default:
        break;
    }
    /* Any non-alphanumeric escaped char is itself. */
    if (!((c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))) {
        ps->p++;
        return c;
    }
    return -1;
}

/* Parse a class-type escape (\d \w \s \h \v and negations). Returns 1 and ORs into out. */
static int parse_class_escape(parser *ps, cls_t *out) {
    if (ps->p >= ps->end) return 0;
    cls_t c;
    memset(&c, 0, sizeof c);
    int neg = 0;
    switch (*ps->p) {
    case 'd': cls_digit(&c); break;
    case 'D': cls_digit(&c); neg = 1; break;
    case 'w': cls_word(&c); break;
    case 'W': cls_word(&c); neg = 1; break;
    case 's': cls_space(&c); break;
    case 'S': cls_space(&c); neg = 1; break;
    case 'h': cls_hspace(&c); break;
    case 'H': cls_hspace(&c); neg = 1; break;
    case 'v': cls_vspace(&c); break;
    case 'V': cls_vspace(&c); neg = 1; break;
    case 'N': cls_set(&c, '\n'); neg = 1; break;
    default: return 0;
    }
    ps->p++;
    if (neg) cls_not(&c);
    cls_or(out, &c);
    return 1;
}

static int posix_class(const char *name, size_t n, cls_t *c) {
#define IS(s) (n == strlen(s) && memcmp(name, s, n) == 0)
    if (IS("alpha")) { cls_range(c, 'a', 'z'); cls_range(c, 'A', 'Z'); }
    else if (IS("digit")) cls_digit(c);
    else if (IS("alnum")) { cls_range(c, 'a', 'z'); cls_range(c, 'A', 'Z'); cls_digit(c); }
    else if (IS("upper")) cls_range(c, 'A', 'Z');
    else if (IS("lower")) cls_range(c, 'a', 'z');
    else if (IS("space")) cls_space(c);
    else if (IS("blank")) { cls_set(c, ' '); cls_set(c, 9); }
    else if (IS("punct")) { for (unsigned b = 33; b < 127; b++) if (!is_word(b) || b == '_') cls_set(c, b); }
    else if (IS("print")) cls_range(c, 32, 126);
    else if (IS("graph")) cls_range(c, 33, 126);
    else if (IS("cntrl")) { cls_range(c, 0, 31); cls_set(c, 127); }
    else if (IS("xdigit")) { cls_digit(c); cls_range(c, 'a', 'f'); cls_range(c, 'A', 'F'); }
    else if (IS("word")) cls_word(c);
    else if (IS("ascii")) cls_range(c, 0, 127);
    else return 0;
#undef IS
    return 1;
}

static node *parse_bracket(parser *ps, unsigned flags) {
    /* ps->p just past '[' */
    cls_t set;
    memset(&set, 0, sizeof set);
    int neg = 0;
    if (ps->p < ps->end && *ps->p == '^') { neg = 1; ps->p++; }
    int first = 1;
    for (;;) {
        if (ps->p >= ps->end) { fail(ps, "unterminated character class"); return NULL; }
        int c = *ps->p;
        if (c == ']' && !first) { ps->p++; break; }
        first = 0;
        int lo = -1;
        if (c == '[' && ps->p + 1 < ps->end && ps->p[1] == ':') {
            const unsigned char *q = ps->p + 2;
            int pneg = 0;
            if (q < ps->end && *q == '^') { pneg = 1; q++; }
            const unsigned char *s = q;
            while (q + 1 < ps->end && !(q[0] == ':' && q[1] == ']')) q++;
            if (q + 1 < ps->end) {
                cls_t pc;
                memset(&pc, 0, sizeof pc);
                if (!posix_class((const char *)s, (size_t)(q - s), &pc)) { fail(ps, "unknown POSIX class"); return NULL; }
                if (pneg) cls_not(&pc);
                cls_or(&set, &pc);
                ps->p = q + 2;
                continue;
            }
        }
        if (c == '\\') {
            ps->p++;
            if (ps->p >= ps->end) { fail(ps, "trailing backslash in class"); return NULL; }
            if (parse_class_escape(ps, &set)) continue;
            lo = parse_byte_escape(ps, 1);
            if (ps->failed) return NULL;
            if (lo < 0) { fail(ps, "unsupported escape in class"); return NULL; }
        } else {
            lo = c;
            ps->p++;
        }
        /* range? */
        if (ps->p + 1 < ps->end && ps->p[0] == '-' && ps->p[1] != ']') {
            const unsigned char *save = ps->p;
            ps->p++;
            int hi;
            if (*ps->p == '\\') {
                ps->p++;
                cls_t tmp;
                memset(&tmp, 0, sizeof tmp);
                const unsigned char *before = ps->p;
                if (parse_class_escape(ps, &tmp)) { /* "a-\d": '-' is literal */
                    ps->p = before;
                    (void)save;
                    cls_set(&set, (unsigned)lo);
                    cls_set(&set, '-');
                    parse_class_escape(ps, &set);
                    continue;
                }
                hi = parse_byte_escape(ps, 1);
                if (ps->failed) return NULL;
                if (hi < 0) { fail(ps, "unsupported escape in class range"); return NULL; }
            } else if (*ps->p == '[' && ps->p + 1 < ps->end && ps->p[1] == ':') {
                fail(ps, "POSIX class as range endpoint");
                return NULL;
            } else {
                hi = *ps->p++;
            }
            if (hi < lo) { fail(ps, "range out of order in character class"); return NULL; }
            cls_range(&set, (unsigned)lo, (unsigned)hi);
        } else {
            cls_set(&set, (unsigned)lo);
        }
    }
    if (flags & F_I) cls_caseless(&set);
    if (neg) cls_not(&set);
    if (cls_empty(&set)) { fail(ps, "empty character class"); return NULL; }
    node *n = mk(N_CLASS);
    n->cls = set;
    return n;
}

static node *parse_alt(parser *ps, unsigned *flags);

static node *byte_node(unsigned b, unsigned flags) {
    node *n = mk(N_CLASS);
    cls_set(&n->cls, b);
    if (flags & F_I) cls_caseless(&n->cls);
    return n;
}

static void skip_extended(parser *ps, unsigned flags) {
    if (!(flags & F_X)) return;
    for (;;) {
        while (ps->p < ps->end && (*ps->p == ' ' || (*ps->p >= 9 && *ps->p <= 13))) ps->p++;
        if (ps->p < ps->end && *ps->p == '#') {
            while (ps->p < ps->end && *ps->p != '\n') ps->p++;
            continue;
        }
        break;
    }
}

/* Parses "(?" constructs; ps->p is just past "(?". Returns node or NULL with *flag_only=1. */
static node *parse_group_ext(parser *ps, unsigned *flags, int *flag_only) {
    *flag_only = 0;
    if (ps->p >= ps->end) { fail(ps, "unterminated group"); return NULL; }
    int c = *ps->p;
    if (c == '#') { /* comment */
        while (ps->p < ps->end && *ps->p != ')') ps->p++;
        if (ps->p >= ps->end) { fail(ps, "unterminated comment"); return NULL; }
        ps->p++;
        *flag_only = 1;
        return NULL;
    }
    if (c == ':') {
        ps->p++;
        unsigned f = *flags;
        node *n = parse_alt(ps, &f);
        if (ps->failed) { free_node(n); return NULL; }
        if (ps->p >= ps->end || *ps->p != ')') { fail(ps, "missing )"); free_node(n); return NULL; }
        ps->p++;
        return n;
    }
    if (c == '=' || c == '!') { fail(ps, "lookahead assertions are not supported"); return NULL; }
    if (c == '>') { fail(ps, "atomic groups are not supported"); return NULL; }
    if (c == '(') { fail(ps, "conditional subpatterns are not supported"); return NULL; }
    if (c == 'R' || c == '&' || c == '+' || (c >= '0' && c <= '9')) { fail(ps, "recursion / subroutine calls are not supported"); return NULL; }
    if (c == 'C') { fail(ps, "callouts are not supported"); return NULL; }
    if (c == '|') { fail(ps, "branch reset groups are not supported"); return NULL; }
    if (c == '<' || c == 'P' || c == '\'') {
        const unsigned char *q = ps->p;
        int close = '>';
        if (c == 'P') {
            q++;
            if (q < ps->end && (*q == '=' || *q == '>')) { fail(ps, "named back-references / recursion are not supported"); return NULL; }
            if (q >= ps->end || *q != '<') { fail(ps, "malformed (?P"); return NULL; }
            q++;
        } else if (c == '<') {
            q++;
            if (q < ps->end && (*q == '=' || *q == '!')) { fail(ps, "lookbehind assertions are not supported"); return NULL; }
        } else {
            q++;
            close = '\'';
        }
        const unsigned char *s = q;
        while (q < ps->end && (is_word(*q))) q++;
        if (q == s || q >= ps->end || *q != close) { fail(ps, "malformed group name"); return NULL; }
        ps->p = q + 1;
        unsigned f = *flags;
        node *n = parse_alt(ps, &f);
        if (ps->failed) { free_node(n); return NULL; }
        if (ps->p >= ps->end || *ps->p != ')') { fail(ps, "missing )"); free_node(n); return NULL; }
        ps->p++;
        return n;
    }
    /* inline flags: (?imsx-imsx) or (?imsx-imsx: ... ) */
    unsigned f = *flags;
    int on = 1, any = 0;
    while (ps->p < ps->end) {
        c = *ps->p;
        unsigned bit = 0;
        if (c == 'i') bit = F_I;
        else if (c == 's') bit = F_S;
        else if (c == 'm') bit = F_M;
        else if (c == 'x') bit = F_X;
        else if (c == '-') { on = 0; ps->p++; any = 1; continue; }
        else break;
        if (on) f |= bit; else f &= ~bit;
        any = 1;
        ps->p++;
    }
    if (!any || ps->p >= ps->end) { fail(ps, "unsupported group construct"); return NULL; }
    if (*ps->p == ')') {
        ps->p++;
        *flags = f;
        *flag_only = 1;
        return NULL;
    }
    if (*ps->p == ':') {
        ps->p++;
        node *n = parse_alt(ps, &f);
        if (ps->failed) { free_node(n); return NULL; }
        if (ps->p >= ps->end || *ps->p != ')') { fail(ps, "missing )"); free_node(n); return NULL; }
        ps->p++;
        return n;
    }
    fail(ps, "unsupported inline flag");
    return NULL;
}

/* Returns an atom node, or NULL when nothing was produced (flag-only group / comment). */
static node *parse_atom(parser *ps, unsigned *flags, int *produced) {
    *produced = 1;
    int c = *ps->p;
    if (c == '(') {
        ps->p++;
        if (++ps->depth > 200) { fail(ps, "pattern nesting too deep"); return NULL; }
        node *n;
        if (ps->p < ps->end && *ps->p == '?') {
            ps->p++;
            int flag_only = 0;
            n = parse_group_ext(ps, flags, &flag_only);
            if (flag_only) *produced = 0;
        } else if (ps->p < ps->end && *ps->p == '*') {
            fail(ps, "backtracking control verbs are not supported");
            n = NULL;
        } else {
            unsigned f = *flags;
            n = parse_alt(ps, &f);
            if (!ps->failed) {
                if (ps->p >= ps->end || *ps->p != ')') { fail(ps, "missing )"); free_node(n); n = NULL; }
                else ps->p++;
            }
        }
        ps->depth--;
        return n;
    }
    if (c == '[') {
        ps->p++;
        return parse_bracket(ps, *flags);
    }
    if (c == '.') {
        ps->p++;
        node *n = mk(N_CLASS);
        cls_range(&n->cls, 0, 255);
        if (!(*flags & F_S)) n->cls.w['\n' >> 5] &= ~(1u << ('\n' & 31));
        return n;
    }
    if (c == '^') {
        ps->p++;
        node *n = mk(N_ASSERT);
        n->akind = (*flags & F_M) ? A_BOL_ML : A_BOL;
        return n;
    }
    if (c == '$') {
        ps->p++;
        node *n = mk(N_ASSERT);
        n->akind = (*flags & F_M) ? A_EOL_ML : A_EOL;
        return n;
    }
    if (c == '\\') {
        ps->p++;
        if (ps->p >= ps->end) { fail(ps, "trailing backslash"); return NULL; }
        int e = *ps->p;
        node *n;
        switch (e) {
        case 'b': ps->p++; n = mk(N_ASSERT); n->akind = A_WB; return n;
        case 'B': ps->p++; n = mk(N_ASSERT); n->akind = A_NWB; return n;
        case 'A': ps->p++; n = mk(N_ASSERT); n->akind = A_BOL; return n;
        case 'Z': ps->p++; n = mk(N_ASSERT); n->akind = A_EOL; return n;
        case 'z': ps->p++; n = mk(N_ASSERT); n->akind = A_EOD; return n;
        case 'Q': {
            ps->p++;
            node *cat = mk(N_CAT);
            while (ps->p < ps->end) {
                if (ps->p + 1 < ps->end && ps->p[0] == '\\' && ps->p[1] == 'E') { ps->p += 2; break; }
                add_kid(cat, byte_node(*ps->p++, *flags));
            }
            if (cat->nkids == 0) { free_node(cat); *produced = 0; return NULL; }
            /* a quantifier after \Q..\E binds to the last literal only: parse_cat splits it off */
            return cat;
        }
        case 'E': ps->p++; *produced = 0; return NULL;
        case 'G': case 'K': case 'X': case 'R': case 'C':
            fail(ps, "\\%c is not supported", e);
            return NULL;
        case 'p': case 'P':
            fail(ps, "unicode properties require UCP, which is not supported");
            return NULL;
        case 'g': case 'k':
            fail(ps, "back-references are not supported");
            return NULL;
        default: break;
        }
        n = mk(N_CLASS);
        if (parse_class_escape(ps, &n->cls)) {
            if (*flags & F_I) cls_caseless(&n->cls);
            return n;
        }
        free_node(n);
        int b = parse_byte_escape(ps, 0);
        if (ps->failed) return NULL;
        if (b < 0) {
            if (e >= '1' && e <= '9') fail(ps, "back-references are not supported");
            else fail(ps, "unsupported escape \\%c", e);
            return NULL;
        }
        return byte_node((unsigned)b, *flags);
    }
    ps->p++;
    return byte_node((unsigned)c, *flags);
}

/* Try to parse {n}, {n,}, {n,m} at ps->p (pointing at '{'). Returns 1 on success. */
static int parse_braces(parser *ps, int *mn, int *mx) {
    const unsigned char *q = ps->p + 1;
    long a = 0, b = -1;
    int nd = 0;
    while (q < ps->end && *q >= '0' && *q <= '9') {
        a = a * 10 + (*q - '0');
        if (a > 100000) a = 100000;
        q++;
        nd++;
    }
    if (nd == 0) return 0;
    if (q < ps->end && *q == '}') {
        b = a;
        q++;
    } else if (q < ps->end && *q == ',') {
        q++;
        nd = 0;
        long v = 0;
        while (q < ps->end && *q >= '0' && *q <= '9') {
            v = v * 10 + (*q - '0');
            if (v > 100000) v = 100000;
            q++;
            nd++;
        }
        if (q >= ps->end || *q != '}') return 0;
        q++;
        b = nd ? v : -1;
    } else {
        return 0;
    }
    ps->p = q;
    *mn = (int)a;
    *mx = (int)b;
    return 1;
}

static node *parse_cat(parser *ps, unsigned *flags) {
    node *cat = mk(N_CAT);
    for (;;) {
        skip_extended(ps, *flags);
        if (ps->p >= ps->end || *ps->p == '|' || *ps->p == ')') break;
        int c = *ps->p;
        if (c == '*' || c == '+' || c == '?') { fail(ps, "nothing to repeat"); break; }
        int produced = 1;
        node *a = parse_atom(ps, flags, &produced);
        if (ps->failed) { free_node(a); break; }
        if (!produced) continue;
        /* \Q..\E sequences: quantifier binds to the last literal only */
        node *prefix = NULL;
        if (a->type == N_CAT && a->nkids > 1 && ps->p < ps->end) {
            /* only \Q..\E and groups produce CAT; groups come wrapped by parse_alt (ALT or CAT). To keep
             * group semantics exact, groups are always returned as N_ALT by parse_alt; a bare N_CAT here
             * is a \Q..\E literal run. */
            node *last = a->kids[a->nkids - 1];
            a->nkids--;
            prefix = a;
            a = last;
        }
        if (prefix) {
            for (int i = 0; i < prefix->nkids; i++) add_kid(cat, prefix->kids[i]);
            prefix->nkids = 0;
            free_node(prefix);
        }
        /* quantifiers (possibly stacked like a{2}{3} — PCRE allows; keep simple: loop) */
        for (;;) {
            skip_extended(ps, *flags);
            if (ps->p >= ps->end) break;
            int q = *ps->p;
            int mn, mx;
            if (q == '*') { mn = 0; mx = -1; ps->p++; }
            else if (q == '+') { mn = 1; mx = -1; ps->p++; }
            else if (q == '?') { mn = 0; mx = 1; ps->p++; }
            else if (q == '{') {
                if (!parse_braces(ps, &mn, &mx)) break; /* literal '{' handled as next atom */
            } else break;
            if (mx >= 0 && mx < mn) { fail(ps, "numbers out of order in {} quantifier"); break; }
            if (mn > 32767 || mx > 32767) { fail(ps, "bounded repeat is too large"); break; }
            if (ps->p < ps->end && *ps->p == '+') { fail(ps, "possessive quantifiers are not supported"); break; }
            if (ps->p < ps->end && *ps->p == '?') ps->p++; /* lazy: same match-end set */
            if (a->type == N_ASSERT) { fail(ps, "quantifier on a zero-width assertion"); break; }
            node *r = mk(N_REP);
            r->min = mn;
            r->max = mx;
            add_kid(r, a);
            a = r;
        }
        if (ps->failed) { free_node(a); break; }
        add_kid(cat, a);
    }
    return cat;
}

static node *parse_alt(parser *ps, unsigned *flags) {
    node *alt = mk(N_ALT);
    for (;;) {
        node *c = parse_cat(ps, flags);
        add_kid(alt, c);
        if (ps->failed) break;
        if (ps->p < ps->end && *ps->p == '|') { ps->p++; continue; }
        break;
    }
    return alt;
}

/* ------------------------------------------------------------------ analysis */
static int nullable(const node *n) {
    switch (n->type) {
    case N_EMPTY: case N_ASSERT: return 1;
    case N_CLASS: return 0;
    case N_CAT:
        for (int i = 0; i < n->nkids; i++) if (!nullable(n->kids[i])) return 0;
        return 1;
    case N_ALT:
        for (int i = 0; i < n->nkids; i++) if (nullable(n->kids[i])) return 1;
        return 0;
    case N_REP: return n->min == 0 || nullable(n->kids[0]);
    }
    return 1;
}
static int can_consume(const node *n) {
    switch (n->type) {
    case N_CLASS: return 1;
    case N_CAT: case N_ALT:
        for (int i = 0; i < n->nkids; i++) if (can_consume(n->kids[i])) return 1;
        return 0;
    case N_REP: return (n->max != 0) && can_consume(n->kids[0]);
    default: return 0;
    }
}
/* Hyperscan rejects start anchors that can be preceded by consumed input, and end anchors that
 * can be followed by consumed input, outside multiline mode ("Embedded start/end anchors not
 * supported").  before/after: whether input may be consumed before/after this node. */
static void check_embedded(parser *ps, const node *n, int before, int after) {
    if (ps->failed) return;
    switch (n->type) {
    case N_ASSERT:
        if (n->akind == A_BOL && before) fail(ps, "embedded start anchors are not supported");
        if ((n->akind == A_EOL || n->akind == A_EOD) && after) fail(ps, "embedded end anchors are not supported");
        break;
    case N_CAT: {
        for (int i = 0; i < n->nkids; i++) {
            int b = before, a = after;
            for (int j = 0; j < i; j++) if (can_consume(n->kids[j])) b = 1;
            for (int j = i + 1; j < n->nkids; j++) if (can_consume(n->kids[j])) a = 1;
            check_embedded(ps, n->kids[i], b, a);
        }
        break;
    }
    case N_ALT:
        for (int i = 0; i < n->nkids; i++) check_embedded(ps, n->kids[i], before, after);
        break;
    case N_REP: {
        int loops = (n->max < 0 || n->max > 1) && can_consume(n->kids[0]);
        check_embedded(ps, n->kids[0], before || loops, after || loops);
        break;
    }
    default: break;
    }
}

/* ------------------------------------------------------------------ program */
enum { I_CLASS, I_SPLIT, I_JMP, I_ASSERT, I_MATCH };
typedef struct {
    int op;
    int x, y;
} inst;

struct orx_prog {
    inst *code;
    int n, cap;
    cls_t *classes;
    int ncls, clscap;
    /* scratch for the simulation (per prog: one prog must not be scanned from two threads at once) */
    int *mark;
    int *l0, *l1, *stk;
    int gen_counter;
};

#define ORX_MAX_INST 400000

static int emit(orx_prog *pr, int op, int x, int y) {
    if (pr->n == pr->cap) {
        pr->cap = pr->cap ? pr->cap * 2 : 64;
        pr->code = (inst *)realloc(pr->code, sizeof(inst) * (size_t)pr->cap);
    }
    pr->code[pr->n].op = op;
    pr->code[pr->n].x = x;
    pr->code[pr->n].y = y;
    return pr->n++;
}
static int add_class(orx_prog *pr, const cls_t *c) {
    for (int i = 0; i < pr->ncls; i++) if (memcmp(&pr->classes[i], c, sizeof *c) == 0) return i;
    if (pr->ncls == pr->clscap) {
        pr->clscap = pr->clscap ? pr->clscap * 2 : 16;
        pr->classes = (cls_t *)realloc(pr->classes, sizeof(cls_t) * (size_t)pr->clscap);
    }
    pr->classes[pr->ncls] = *c;
    return pr->ncls++;
}

static int gen(orx_prog *pr, const node *n);

static int gen(orx_prog *pr, const node *n) {
    if (pr->n > ORX_MAX_INST) return -1;
    switch (n->type) {
    case N_EMPTY: return 0;
    case N_CLASS: emit(pr, I_CLASS, add_class(pr, &n->cls), 0); return 0;
    case N_ASSERT: emit(pr, I_ASSERT, n->akind, 0); return 0;
    case N_CAT:
        for (int i = 0; i < n->nkids; i++) if (gen(pr, n->kids[i])) return -1;
        return 0;
    case N_ALT: {
        if (n->nkids == 1) return gen(pr, n->kids[0]);
        int *jmps = (int *)malloc(sizeof(int) * (size_t)n->nkids);
        int nj = 0;
        for (int i = 0; i < n->nkids; i++) {
            int sp = -1;
            if (i + 1 < n->nkids) sp = emit(pr, I_SPLIT, pr->n + 1, 0);
            if (gen(pr, n->kids[i])) { free(jmps); return -1; }
            if (i + 1 < n->nkids) {
                jmps[nj++] = emit(pr, I_JMP, 0, 0);
                pr->code[sp].y = pr->n;
            }
        }
        for (int i = 0; i < nj; i++) pr->code[jmps[i]].x = pr->n;
        free(jmps);
        return 0;
    }
    case N_REP: {
        const node *k = n->kids[0];
        for (int i = 0; i < n->min; i++) if (gen(pr, k)) return -1;
        if (n->max < 0) {
            /* L1: split L2, L3; L2: k; jmp L1; L3: */
            int l1 = emit(pr, I_SPLIT, pr->n + 1, 0);
            if (gen(pr, k)) return -1;
            emit(pr, I_JMP, l1, 0);
            pr->code[l1].y = pr->n;
        } else {
            int extra = n->max - n->min;
            int *sps = (int *)malloc(sizeof(int) * (size_t)(extra > 0 ? extra : 1));
            for (int i = 0; i < extra; i++) {
                sps[i] = emit(pr, I_SPLIT, pr->n + 1, 0);
                if (gen(pr, k)) { free(sps); return -1; }
            }
            for (int i = 0; i < extra; i++) pr->code[sps[i]].y = pr->n;
            free(sps);
        }
        return 0;
    }
    }
    return -1;
}

orx_prog *orx_compile(const char *pattern, unsigned flags, char *err, size_t errlen) {
    parser ps;
    memset(&ps, 0, sizeof ps);
    ps.p = (const unsigned char *)pattern;
    ps.end = ps.p + strlen(pattern);
    ps.err = err;
    ps.errlen = errlen;
    if (err && errlen) err[0] = 0;
    if (flags & ~ORX_FLAGS_SUPPORTED) {
        fail(&ps, "unsupported flag bits 0x%x", flags & ~ORX_FLAGS_SUPPORTED);
        return NULL;
    }
    if (ps.p == ps.end) {
        fail(&ps, "empty pattern");
        return NULL;
    }
    unsigned f = 0;
    if (flags & ORX_FLAG_CASELESS) f |= F_I;
    if (flags & ORX_FLAG_DOTALL) f |= F_S;
    if (flags & ORX_FLAG_MULTILINE) f |= F_M;
    node *root = parse_alt(&ps, &f);
    if (!ps.failed && ps.p < ps.end) fail(&ps, "unmatched )");
    if (!ps.failed && nullable(root)) fail(&ps, "pattern can match the empty string (HS_FLAG_ALLOWEMPTY not supported)");
    if (!ps.failed) check_embedded(&ps, root, 0, 0);
    if (ps.failed) {
        free_node(root);
        return NULL;
    }
    orx_prog *pr = (orx_prog *)calloc(1, sizeof *pr);
    if (gen(pr, root) || pr->n > ORX_MAX_INST) {
        fail(&ps, "pattern too large");
        free_node(root);
        orx_free(pr);
        return NULL;
    }
    emit(pr, I_MATCH, 0, 0);
    free_node(root);
    pr->mark = (int *)calloc((size_t)pr->n, sizeof(int));
    pr->l0 = (int *)malloc(sizeof(int) * (size_t)pr->n);
    pr->l1 = (int *)malloc(sizeof(int) * (size_t)pr->n);
    pr->stk = (int *)malloc(sizeof(int) * (size_t)pr->n * 2 + 16);
    return pr;
}

void orx_free(orx_prog *p) {
    if (!p) return;
    free(p->code);
    free(p->classes);
    free(p->mark);
    free(p->l0);
    free(p->l1);
    free(p->stk);
    free(p);
}

/* ------------------------------------------------------------------ simulation */
static int assert_holds(int kind, const unsigned char *d, size_t len, size_t i) {
    int has_prev = i > 0, has_next = i < len;
    unsigned prev = has_prev ? d[i - 1] : 0, next = has_next ? d[i] : 0;
    switch (kind) {
    case A_BOL_ML: return !has_prev || prev == '\n';
    case A_BOL: return !has_prev;
    case A_EOL_ML: return !has_next || next == '\n';
    case A_EOL: return !has_next || (i == len - 1 && next == '\n');
    case A_EOD: return !has_next;
    case A_WB: return (has_prev && is_word(prev)) != (has_next && is_word(next));
    case A_NWB: return (has_prev && is_word(prev)) == (has_next && is_word(next));
    }
    return 0;
}

size_t orx_scan(const orx_prog *pc, const unsigned char *data, size_t len, int single, orx_report_fn cb, void *ctx) {
    orx_prog *p = (orx_prog *)pc; /* scratch is per-prog: not re-entrant, fine for test use */
    int *clist = p->l0, *pending = p->l1;
    int npend = 0;
    size_t reports = 0;
    for (size_t i = 0; i <= len; i++) {
        /* closure of pending ∪ {0} at position i */
        if (p->gen_counter == 0x7ffffffe) {
            memset(p->mark, 0, sizeof(int) * (size_t)p->n);
            p->gen_counter = 0;
        }
        int gen_id = ++p->gen_counter;
        int nc = 0, matched = 0, sp = 0;
        for (int k = npend; k >= 0; k--) {
            p->stk[sp++] = (k == npend) ? 0 : pending[k];
            while (sp) {
                int pcx = p->stk[--sp];
                if (p->mark[pcx] == gen_id) continue;
                p->mark[pcx] = gen_id;
                const inst *in = &p->code[pcx];
                switch (in->op) {
                case I_CLASS: clist[nc++] = pcx; break;
                case I_SPLIT: p->stk[sp++] = in->y; p->stk[sp++] = in->x; break;
                case I_JMP: p->stk[sp++] = in->x; break;
                case I_ASSERT:
                    if (assert_holds(in->x, data, len, i)) p->stk[sp++] = pcx + 1;
                    break;
                case I_MATCH: matched = 1; break;
                }
            }
        }
        if (matched) {
            reports++;
            if (cb && cb(i, ctx)) return reports;
            if (single) return reports;
        }
        if (i == len) break;
        unsigned b = data[i];
        npend = 0;
        for (int k = 0; k < nc; k++) {
            const inst *in = &p->code[clist[k]];
            if (cls_has(&p->classes[in->x], b)) pending[npend++] = clist[k] + 1;
        }
    }
    return reports;
}
