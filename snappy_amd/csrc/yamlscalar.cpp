// yamlscalar.cpp -- how yaml.v2 renders a Go string as the value of `name:` in hashes.yaml.
//
// writeHashes marshals every file name (snappy/build.go:249-264: `Name: path[len(buildDir)+1:]`, then
// yaml.Marshal(hashes)).  The bytes come from gopkg.in/yaml.v2 @ 49c95bdc (dependencies.tsv:7), which is NOT in the
// reference tree: this file restates its published algorithm --
//   encode.go  stringv():  double-quoted when the text would resolve to another type (resolve.go: bool/null/int/float,
//                          base-60 floats), else plain;
//   emitterc.go (the libyaml port)  yaml_emitter_analyze_scalar / select_scalar_style: plain -> single-quoted when the
//                          text is not allowed as a plain scalar in block context, -> double-quoted when it holds
//                          characters single quotes cannot carry;
//   the three scalar writers, with their folding of long lines at spaces (best_width 80; a folded line is indented by 4 inside a list item).
// PARITY UNPINNED: the reference's tests hold plain names only (snappy/hashes_test.go:89-103); nothing in them covers
// a quoted or folded name.  tests/test_yaml_names.py checks this restatement against two other readers/writers of the
// format (the repository's own parser and PyYAML), which is a round-trip and a second opinion, not parity.
// What would depend on the Go version the reference was built with, or on a !!binary tag, is refused (SNAPHASH_ENAME).
#include "hostpass.h"

#include <string.h>

namespace snaphash {

namespace {

bool is_digit(unsigned char c) { return c >= '0' && c <= '9'; }
bool is_hex(unsigned char c) { return is_digit(c) || (c >= 'a' && c <= 'f') || (c >= 'A' && c <= 'F'); }

bool eq_nocase(const std::string& s, const char* t)
{
    const size_t n = strlen(t);
    if (s.size() != n) return false;
    for (size_t i = 0; i < n; ++i) {
        unsigned char a = (unsigned char)s[i], b = (unsigned char)t[i];
        if (a >= 'A' && a <= 'Z') a = (unsigned char)(a - 'A' + 'a');
        if (a != b) return false;
    }
    return true;
}

// strconv.ParseFloat's decimal grammar (Go 1.3/1.4): [+-] digits [. digits] | . digits, optional exponent; plus the
// special spellings.  Hexadecimal floats (Go >= 1.13) are answered by `ambiguous` below, not here.
bool go_parse_float_ok(const std::string& s)
{
    size_t i = 0;
    const size_t n = s.size();
    if (n == 0) return false;
    if (s[0] == '+' || s[0] == '-') {
        const std::string rest = s.substr(1);
        if (eq_nocase(rest, "inf") || eq_nocase(rest, "infinity")) return true;
        i = 1;
    } else if (eq_nocase(s, "inf") || eq_nocase(s, "infinity") || eq_nocase(s, "nan")) {
        return true;
    }
    size_t nd = 0;
    while (i < n && is_digit((unsigned char)s[i])) { ++i; ++nd; }
    if (i < n && s[i] == '.') {
        ++i;
        while (i < n && is_digit((unsigned char)s[i])) { ++i; ++nd; }
    }
    if (nd == 0) return false;
    if (i < n && (s[i] == 'e' || s[i] == 'E')) {
        ++i;
        if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
        size_t ne = 0;
        while (i < n && is_digit((unsigned char)s[i])) { ++i; ++ne; }
        if (ne == 0) return false;
    }
    return i == n;
}

// ---- what strconv answers with err == nil -----------------------------------------------------------------------
// resolve.go (yaml.v2 @ 49c95bdc) takes a scalar for a number only when the strconv call returns NO error, and a value
// out of range is an error (ErrRange): "1e999", a 0x literal of more than 64 bits or 65 binary digits stay strings and
// are written plain.

// magnitude of a run of digits in `base`; false = a character that is no digit of that base (or no digit at all)
bool parse_magnitude(const std::string& s, size_t i, unsigned base, uint64_t* mag, bool* overflow)
{
    if (i >= s.size()) return false;
    unsigned __int128 v = 0;
    *overflow = false;
    for (; i < s.size(); ++i) {
        const unsigned char c = (unsigned char)s[i];
        unsigned d;
        if (is_digit(c)) d = c - '0';
        else if (c >= 'a' && c <= 'f') d = c - 'a' + 10;
        else if (c >= 'A' && c <= 'F') d = c - 'A' + 10;
        else return false;
        if (d >= base) return false;
        v = v * base + d;
        if (v >> 64) { *overflow = true; v &= ~(unsigned __int128)0 >> 64; } // keep scanning: a bad digit further on is a syntax error, not a range error
    }
    *mag = (uint64_t)v;
    return true;
}

// strconv.ParseInt(s, base, 64) == nil or strconv.ParseUint(s, base, 64) == nil, as resolve.go tries them in turn.
// base 0 (Go before 1.13): "0x" hex, a leading "0" octal, decimal otherwise.  ParseUint takes no sign.
bool go_int_ok(const std::string& s, unsigned base, bool try_unsigned)
{
    size_t i = 0;
    const bool neg = !s.empty() && s[0] == '-';
    const bool sign = !s.empty() && (s[0] == '+' || s[0] == '-');
    if (sign) i = 1;
    if (i >= s.size()) return false;
    if (base == 0) {
        if (s[i] == '0' && i + 1 < s.size() && (s[i + 1] == 'x' || s[i + 1] == 'X')) { base = 16; i += 2; }
        else if (s[i] == '0') base = 8;
        else base = 10;
    }
    uint64_t mag = 0;
    bool overflow = false;
    if (!parse_magnitude(s, i, base, &mag, &overflow)) return false;
    if (overflow) return false;
    if (neg ? mag <= (1ull << 63) : mag <= (1ull << 63) - 1) return true; // ParseInt
    return try_unsigned && !sign;                                         // ParseUint
}

// A decimal that strconv.ParseFloat (correctly rounded, like every IEEE strtod) turns into +-Inf: ErrRange.  Exact and
// locale-free: the digits against 2^1024 - 2^970, the midpoint between the largest double and 2^1024 (a tie rounds to
// even, which is up).  s has passed go_parse_float_ok's decimal grammar.  Underflow to zero is no error in Go.
bool decimal_float_overflows(const std::string& s)
{
    static const char kLimit[] = // 2^1024 - 2^970, 309 digits
        "17976931348623158079372897140530341507993413271003782693617377898044496829276475094664901797758720709633028641669288"
        "79109465555478519404026306574886715058206819089020007083836762738548458177115317644757302700698555713669596228429148"
        "19860834936475292719074168444365510704342711559699508093042880177904174497792";
    size_t i = 0;
    const size_t n = s.size();
    if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
    std::string digits;
    long point = 0;
    for (; i < n && is_digit((unsigned char)s[i]); ++i) { digits += s[i]; ++point; }
    if (i < n && s[i] == '.') for (++i; i < n && is_digit((unsigned char)s[i]); ++i) digits += s[i];
    long exp10 = 0;
    if (i < n && (s[i] == 'e' || s[i] == 'E')) {
        ++i;
        bool eneg = false;
        if (i < n && (s[i] == '+' || s[i] == '-')) { eneg = s[i] == '-'; ++i; }
        for (; i < n && is_digit((unsigned char)s[i]); ++i) if (exp10 < 100000000) exp10 = exp10 * 10 + (s[i] - '0');
        if (eneg) exp10 = -exp10;
    }
    size_t lead = 0;
    while (lead < digits.size() && digits[lead] == '0') ++lead;
    if (lead == digits.size()) return false; // zero
    digits.erase(0, lead);
    const long mag10 = point - (long)lead + exp10; // value = 0.digits x 10^mag10
    if (mag10 != 309) return mag10 > 309;
    for (size_t k = 0; k < 309; ++k) { // digits, padded with zeros, against the limit
        const char d = k < digits.size() ? digits[k] : '0';
        if (d != kLimit[k]) return d > kLimit[k];
    }
    return true; // equal up to 309 digits: at or above the midpoint
}

bool go_float_ok(const std::string& s)
{
    if (!go_parse_float_ok(s)) return false;
    const size_t i = (s[0] == '+' || s[0] == '-') ? 1 : 0;
    if (i < s.size() && !is_digit((unsigned char)s[i]) && s[i] != '.') return true; // inf, infinity, nan
    return !decimal_float_overflows(s);
}

// resolve.go: binary integers written 0b / -0b (yaml.v2 handles them itself, before Go's strconv learnt the prefix):
// ParseInt(plain[2:], 2, 64), then ParseUint; behind "-0b" ParseInt(plain[3:], 2, 64) alone.  strconv takes a sign
// there too ("0b-101" is -5).
bool yaml_binary_int_ok(const std::string& s)
{
    if (s.compare(0, 2, "0b") == 0) return go_int_ok(s.substr(2), 2, true);
    if (s.compare(0, 3, "-0b") == 0) return go_int_ok(s.substr(3), 2, false);
    return false;
}

// the spelling alone, whatever the range (go_version_dependent below asks)
bool yaml_binary_int(const std::string& s)
{
    size_t i = (s.size() > 0 && s[0] == '-') ? 1 : 0;
    if (s.size() < i + 3 || s[i] != '0' || s[i + 1] != 'b') return false;
    for (i += 2; i < s.size(); ++i) if (s[i] != '0' && s[i] != '1') return false;
    return true;
}

// Spellings whose type depends on the strconv of the Go release that built the reference (unpinned, debian/control:11):
// 0o octal integers and hexadecimal floats parse from Go 1.13 on and are strings before.
bool go_version_dependent(const std::string& plain)
{
    size_t i = 0;
    if (i < plain.size() && (plain[i] == '+' || plain[i] == '-')) ++i;
    if (plain.size() < i + 3 || plain[i] != '0') return false;
    const char p = plain[i + 1];
    if (p == 'o' || p == 'O') {
        for (size_t k = i + 2; k < plain.size(); ++k) if (plain[k] < '0' || plain[k] > '7') return false;
        return true;
    }
    if (p == 'x' || p == 'X') {
        bool has_p = false;
        for (size_t k = i + 2; k < plain.size(); ++k) {
            const unsigned char c = (unsigned char)plain[k];
            if (c == 'p' || c == 'P') has_p = true;
            else if (!is_hex(c) && c != '.' && c != '+' && c != '-') return false;
        }
        return has_p;
    }
    if (p == 'b' || p == 'B') { // "+0b1", "0B1": yaml.v2's own rule covers only 0b / -0b
        if (yaml_binary_int(plain)) return false;
        for (size_t k = i + 2; k < plain.size(); ++k) if (plain[k] != '0' && plain[k] != '1') return false;
        return true;
    }
    return false;
}

// encode.go isBase60Float: ^[-+]?[0-9][0-9_]*(?::[0-5]?[0-9])+(?:\.[0-9_]*)?$
bool is_base60_float(const std::string& s)
{
    size_t i = 0;
    const size_t n = s.size();
    if (n == 0) return false;
    const unsigned char c0 = (unsigned char)s[0];
    if (!(c0 == '+' || c0 == '-' || is_digit(c0)) || s.find(':') == std::string::npos) return false;
    if (s[i] == '+' || s[i] == '-') ++i;
    if (i >= n || !is_digit((unsigned char)s[i])) return false;
    ++i;
    while (i < n && (is_digit((unsigned char)s[i]) || s[i] == '_')) ++i;
    size_t groups = 0;
    while (i < n && s[i] == ':') {
        ++i;
        if (i < n && s[i] >= '0' && s[i] <= '5' && i + 1 < n && is_digit((unsigned char)s[i + 1])) i += 2;
        else if (i < n && is_digit((unsigned char)s[i])) i += 1;
        else return false;
        ++groups;
    }
    if (!groups) return false;
    if (i < n && s[i] == '.') {
        ++i;
        while (i < n && (is_digit((unsigned char)s[i]) || s[i] == '_')) ++i;
    }
    return i == n;
}

// resolve.go resolve("", s) for a valid UTF-8 string: 1 = another type than !!str (encode.go then writes it
// double-quoted), 0 = a string, -1 = depends on the Go version / not restated.
int resolves_to_non_string(const std::string& s)
{
    if (s.empty()) return 1; // "" is in the map (null)
    const unsigned char c0 = (unsigned char)s[0];
    const bool hint_m = strchr("yYnNtTfFoO~", c0) != nullptr;
    const bool hint_num = c0 == '+' || c0 == '-' || is_digit(c0);
    const bool hint_dot = c0 == '.';
    if (!hint_m && !hint_num && !hint_dot) return s == "<<" ? -1 : 0; // "<<" sits in the map behind a hint that may never fire
    static const char* const mapped[] = {"y", "Y", "yes", "Yes", "YES", "on", "On", "ON", "n", "N", "no", "No", "NO", "off",
                                         "Off", "OFF", "true", "True", "TRUE", "false", "False", "FALSE", "~", "null", "Null",
                                         "NULL", ".nan", ".NaN", ".NAN", ".inf", ".Inf", ".INF", "+.inf", "+.Inf", "+.INF",
                                         "-.inf", "-.Inf", "-.INF", nullptr};
    for (int i = 0; mapped[i]; ++i)
        if (s == mapped[i]) return 1;
    if (hint_m) return 0;
    if (hint_dot) return go_float_ok(s) ? 1 : 0;
    std::string plain; // strings.Replace(in, "_", "", -1)
    for (char c : s) if (c != '_') plain += c;
    if (go_version_dependent(plain)) return -1;
    if (go_int_ok(plain, 0, true) || go_float_ok(plain) || yaml_binary_int_ok(plain)) return 1;
    return 0;
}

// ---- UTF-8 ----------------------------------------------------------------------------------------------------
// length of the well-formed sequence at s[i], 0 if it is not one (utf8.ValidString: no overlongs, no surrogates)
int utf8_len(const std::string& s, size_t i, uint32_t* cp)
{
    const unsigned char c = (unsigned char)s[i];
    auto cont = [&](size_t k) { return i + k < s.size() && ((unsigned char)s[i + k] & 0xC0) == 0x80; };
    if (c < 0x80) { *cp = c; return 1; }
    if (c >= 0xC2 && c <= 0xDF && cont(1)) { *cp = ((c & 0x1Fu) << 6) | ((unsigned char)s[i + 1] & 0x3Fu); return 2; }
    if (c >= 0xE0 && c <= 0xEF && cont(1) && cont(2)) {
        const uint32_t v = ((c & 0x0Fu) << 12) | (((unsigned char)s[i + 1] & 0x3Fu) << 6) | ((unsigned char)s[i + 2] & 0x3Fu);
        if (v < 0x800 || (v >= 0xD800 && v <= 0xDFFF)) return 0;
        *cp = v;
        return 3;
    }
    if (c >= 0xF0 && c <= 0xF4 && cont(1) && cont(2) && cont(3)) {
        const uint32_t v = ((c & 0x07u) << 18) | (((unsigned char)s[i + 1] & 0x3Fu) << 12) | (((unsigned char)s[i + 2] & 0x3Fu) << 6) |
                           ((unsigned char)s[i + 3] & 0x3Fu);
        if (v < 0x10000 || v > 0x10FFFF) return 0;
        *cp = v;
        return 4;
    }
    return 0;
}

// yamlprivateh.go is_printable, by code point: #x0A, #x20-#x7E, #x85?? no: the Go port's byte tests admit
// #xA0-#xD7FF, #xE000-#xFFFD except #xFEFF; NEL (#x85) and everything above #xFFFF are NOT printable there.
bool is_printable_cp(uint32_t cp)
{
    if (cp == 0x0A) return true;
    if (cp >= 0x20 && cp <= 0x7E) return true;
    if (cp >= 0xA0 && cp <= 0xD7FF) return true;
    if (cp >= 0xE000 && cp <= 0xFFFD && cp != 0xFEFF) return true;
    return false;
}

struct Emit { // the emitter state the three scalar writers touch
    std::string& out;
    int column;
    int indent;
    void put(char c) { out += c; ++column; }
    void put_char(const std::string& s, size_t i, int w) { out.append(s, i, (size_t)w); ++column; } // one character = one column
    void write_indent() // emitterc.go yaml_emitter_write_indent, mid-scalar: always a break, then `indent` spaces
    {
        out += '\n';
        column = 0;
        while (column < indent) put(' ');
    }
};

constexpr int kBestWidth = 80; // yaml_emitter_emit_stream_start: an unset best_width becomes 80

} // namespace

// Appends " <scalar>" -- the bytes yaml.v2 writes behind "name:" -- to out.  column: the column behind the ':'
// (7 for "- name:"); indent: what a folded line is indented by -- emitterc.go yaml_emitter_emit_scalar raises the
// indentation by best_indent around the scalar, so inside a list item's mapping (keys at 2) that is 4.
// SNAPHASH_ENAME for a name this restatement does not cover.
int yaml_append_name_scalar(const std::string& s, int column, int indent, std::string& out)
{
    if (s.empty() || s.size() > 4096) return SNAPHASH_ENAME;
    // utf8.ValidString, else yaml.v2 writes !!binary base64: not restated
    std::vector<uint32_t> cps;
    std::vector<uint8_t> widths;
    for (size_t i = 0; i < s.size();) {
        uint32_t cp = 0;
        const int w = utf8_len(s, i, &cp);
        if (w == 0) return SNAPHASH_ENAME;
        if (cp == '\n' || cp == '\r' || cp == 0x85 || cp == 0x2028 || cp == 0x2029) return SNAPHASH_ENAME; // line breaks: literal style / break analysis, not restated
        cps.push_back(cp);
        widths.push_back((uint8_t)w);
        i += (size_t)w;
    }
    const int res = resolves_to_non_string(s);
    if (res < 0) return SNAPHASH_ENAME;
    enum { PLAIN, SINGLE, DOUBLE } style = (res == 1 || is_base60_float(s)) ? DOUBLE : PLAIN;

    // ---- yaml_emitter_analyze_scalar (no line breaks can occur here) ----
    bool block_indicators = false, special = false, leading_space = false, trailing_space = false;
    if (s.size() >= 3 && (s.compare(0, 3, "---") == 0 || s.compare(0, 3, "...") == 0)) block_indicators = true;
    bool preceded_by_ws = true;
    const size_t n = cps.size();
    for (size_t k = 0; k < n; ++k) {
        const uint32_t c = cps[k];
        const bool followed_by_ws = k + 1 >= n || cps[k + 1] == ' ' || cps[k + 1] == '\t';
        if (k == 0) {
            if (c && c < 0x80 && strchr("#,[]{}&*!|>'\"%@`", (int)c)) block_indicators = true;
            else if ((c == '?' || c == ':' || c == '-') && followed_by_ws) block_indicators = true;
        } else {
            if (c == ':' && followed_by_ws) block_indicators = true;
            else if (c == '#' && preceded_by_ws) block_indicators = true;
        }
        if (!is_printable_cp(c)) special = true; // the emitter runs with unicode = true (encode.go newEncoder)
        if (c == ' ') {
            if (k == 0) leading_space = true;
            if (k + 1 == n) trailing_space = true;
        }
        preceded_by_ws = c == ' ' || c == '\t';
    }
    const bool block_plain_allowed = !(leading_space || trailing_space || special || block_indicators);
    const bool single_quoted_allowed = !special;
    // ---- yaml_emitter_select_scalar_style: block context, not a simple key ----
    if (style == PLAIN && !block_plain_allowed) style = SINGLE;
    if (style == SINGLE && !single_quoted_allowed) style = DOUBLE;

    Emit e{out, column, indent};
    size_t bi = 0; // byte index of character k
    if (style == PLAIN) { // yaml_emitter_write_plain_scalar(allow_breaks = true)
        e.put(' ');
        bool spaces = false;
        for (size_t k = 0; k < n; bi += widths[k], ++k) {
            if (cps[k] == ' ') {
                if (!spaces && e.column > kBestWidth && !(k + 1 < n && cps[k + 1] == ' ')) e.write_indent();
                else e.put(' ');
                spaces = true;
            } else {
                e.put_char(s, bi, widths[k]);
                spaces = false;
            }
        }
        return SNAPHASH_OK;
    }
    if (style == SINGLE) { // yaml_emitter_write_single_quoted_scalar
        e.put(' ');
        e.put('\'');
        bool spaces = false;
        for (size_t k = 0; k < n; bi += widths[k], ++k) {
            if (cps[k] == ' ') {
                if (!spaces && e.column > kBestWidth && k > 0 && k + 1 < n && cps[k + 1] != ' ') e.write_indent();
                else e.put(' ');
                spaces = true;
            } else {
                if (cps[k] == '\'') e.put('\'');
                e.put_char(s, bi, widths[k]);
                spaces = false;
            }
        }
        e.put('\'');
        return SNAPHASH_OK;
    }
    // yaml_emitter_write_double_quoted_scalar
    e.put(' ');
    e.put('"');
    bool spaces = false;
    for (size_t k = 0; k < n; bi += widths[k], ++k) {
        const uint32_t c = cps[k];
        if (!is_printable_cp(c) || c == 0xFEFF || c == '"' || c == '\\') {
            e.put('\\');
            switch (c) {
            case 0x00: e.put('0'); break;
            case 0x07: e.put('a'); break;
            case 0x08: e.put('b'); break;
            case 0x09: e.put('t'); break;
            case 0x0B: e.put('v'); break;
            case 0x0C: e.put('f'); break;
            case 0x0D: e.put('r'); break;
            case 0x1B: e.put('e'); break;
            case 0x22: e.put('"'); break;
            case 0x5C: e.put('\\'); break;
            case 0xA0: e.put('_'); break;
            default: {
                int digits;
                if (c <= 0xFF) { e.put('x'); digits = 2; }
                else if (c <= 0xFFFF) { e.put('u'); digits = 4; }
                else { e.put('U'); digits = 8; }
                for (int sh = (digits - 1) * 4; sh >= 0; sh -= 4) e.put("0123456789ABCDEF"[(c >> sh) & 15]);
            }
            }
            spaces = false;
        } else if (c == ' ') {
            if (!spaces && e.column > kBestWidth && k > 0 && k + 1 < n) {
                e.write_indent();
                if (cps[k + 1] == ' ') e.put('\\');
            } else {
                e.put(' ');
            }
            spaces = true;
        } else {
            e.put_char(s, bi, widths[k]);
            spaces = false;
        }
    }
    e.put('"');
    return SNAPHASH_OK;
}

bool name_emittable(const std::string& s)
{
    if (plain_safe_name(s)) return true; // the pinned case, as the emitter short-cuts it (hostpass.cpp emit_yaml)
    std::string tmp;
    return yaml_append_name_scalar(s, 7, 4, tmp) == SNAPHASH_OK;
}

} // namespace snaphash
