// The reference's `info` and `formats` columns when they are NOT parsed into structs (exon.vcf_parse_info / _formats = false,
// the default schema: schema_builder.rs:119-121, both Utf8).  They are not the fields' bytes: LazyVCFArrayBuilder walks the
// parsed entries and prints them again (exon-vcf/src/array_builder/lazy_array_builder.rs:216-297 and :310-423) --
//   info     key=value joined by ';'; a Flag prints "key=true"; Integer / Float values go through Rust's Display of i32 / f32
//            ("0.50" -> "0.5", "1e-5" -> "0.00001", "1.0" -> "1", "007" -> "7"); list items are joined by ',', a missing item
//            prints "."; INFO "." (no entries) prints the empty string; a missing VALUE ("key=.") is `value_option.unwrap()` on
//            None: the reference panics, this reader reports an error.
//   formats  the FORMAT keys joined by ':', a TAB, then the samples joined by TABs; every sample's values printed like the
//            INFO values and joined by ':'; a genotype prints its alleles with the separator in front of allele i taken from
//            allele i - 1's phasing (:331-360); missing items of a Character list are skipped (:364-368), of the others "."; a
//            missing value panics like INFO's; a record without samples prints "\t".
// Types come from the header's ##INFO / ##FORMAT lines, else from the reserved keys of the VCF specification (noodles falls
// back to them, then to String).  noodles itself is not in /root/reference (Cargo.lock: noodles-vcf 0.70): parity is pinned
// on slt/vcf-select-tests.slt:6-16 (index.vcf) and otherwise follows the reference's printing code line by line; two things
// rest on noodles' behaviour as published and are marked below (the first allele's phasing before VCF 4.4; no
// percent-decoding is applied to String values here).
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace exon {

// Rust's `{}` of an f32: the shortest digits that round-trip, never an exponent, "NaN" / "inf" / "-inf", "-0" for negative zero
inline void rust_f32_display(float v, std::string* out) {
  if (std::isnan(v)) { out->append("NaN"); return; }
  if (std::isinf(v)) { out->append(v < 0 ? "-inf" : "inf"); return; }
  char b[48];
  const auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::scientific);  // shortest round-trip digits: d[.ddd]e+XX
  const char* p = b;
  if (*p == '-') { out->push_back('-'); ++p; }
  char digits[16];
  int nd = 0;
  for (; p < r.ptr && *p != 'e'; ++p)
    if (*p != '.') digits[nd++] = *p;
  int e = 0;
  if (p < r.ptr) {
    ++p;
    const bool neg = *p == '-';
    if (*p == '-' || *p == '+') ++p;
    for (; p < r.ptr; ++p) e = e * 10 + (*p - '0');
    if (neg) e = -e;
  }
  if (e >= nd - 1) {  // an integer: digits, then zeros
    out->append(digits, (size_t)nd);
    out->append((size_t)(e - (nd - 1)), '0');
  } else if (e >= 0) {
    out->append(digits, (size_t)e + 1);
    out->push_back('.');
    out->append(digits + e + 1, (size_t)(nd - e - 1));
  } else {
    out->append("0.");
    out->append((size_t)(-e - 1), '0');
    out->append(digits, (size_t)nd);
  }
}

// value types: 'i' Integer, 'f' Float, 'b' Flag, 'c' Character, 's' String, 'g' the GT key of FORMAT
struct VcfKeyTypes {
  std::unordered_map<std::string, char> info, format;
  static char type_char(const std::string& ty) {
    if (ty == "Integer") return 'i';
    if (ty == "Float") return 'f';
    if (ty == "Flag") return 'b';
    if (ty == "Character") return 'c';
    return 's';
  }
  // reserved keys (VCF 4.3 tables 1 and 2 + the structural-variant keys): what noodles takes when the header has no line
  static char reserved_info(const std::string& k) {
    static const char* const ints[] = {"AC", "AD", "ADF", "ADR", "AN", "DP", "END", "MQ0", "NS", "SB", "SVLEN", "CIPOS", "CIEND", "HOMLEN",
                                       "CILEN", "DPADJ", "CN", "CNADJ", "CICN", "CICNADJ"};
    static const char* const floats[] = {"AF", "BQ", "MQ"};
    static const char* const flags[] = {"DB", "H2", "H3", "SOMATIC", "VALIDATED", "1000G", "IMPRECISE", "NOVEL"};
    for (const char* s : ints) if (k == s) return 'i';
    for (const char* s : floats) if (k == s) return 'f';
    for (const char* s : flags) if (k == s) return 'b';
    return 's';
  }
  static char reserved_format(const std::string& k) {
    static const char* const ints[] = {"AD", "ADF", "ADR", "DP", "EC", "GQ", "HQ", "MQ", "PL", "PP", "PQ", "PS", "CN", "NQ", "HAP", "AHAP"};
    static const char* const floats[] = {"GL", "GP", "CNQ", "CNL", "CNP"};
    if (k == "GT") return 'g';
    for (const char* s : ints) if (k == s) return 'i';
    for (const char* s : floats) if (k == s) return 'f';
    return 's';
  }
  char info_type(const char* k, size_t n) const {
    const std::string key(k, n);
    const auto it = info.find(key);
    return it != info.end() ? it->second : reserved_info(key);
  }
  char format_type(const char* k, size_t n) const {
    const std::string key(k, n);
    if (key == "GT") return 'g';
    const auto it = format.find(key);
    return it != format.end() ? it->second : reserved_format(key);
  }
};

namespace vcf_text_detail {
inline void print_i32(const char* p, size_t n, std::string* out) {
  size_t i = 0;
  bool neg = false;
  if (i < n && (p[i] == '-' || p[i] == '+')) neg = p[i++] == '-';
  if (i == n) throw std::runtime_error("invalid integer '" + std::string(p, n) + "'");
  int64_t v = 0;
  for (; i < n; ++i) {
    if (p[i] < '0' || p[i] > '9') throw std::runtime_error("invalid integer '" + std::string(p, n) + "'");
    v = v * 10 + (p[i] - '0');
    if (v > (int64_t)INT32_MAX + 1) throw std::runtime_error("integer out of the int32 range '" + std::string(p, n) + "'");
  }
  if (neg) v = -v;
  if (v > INT32_MAX) throw std::runtime_error("integer out of the int32 range '" + std::string(p, n) + "'");
  out->append(std::to_string(v));
}
// one value of type `ty`: items split on ',', "." items kept (skipped for Character lists with more than one item)
template <typename ParseF32>
inline void print_value(char ty, const char* v, size_t n, ParseF32 parse_f32, std::string* out) {
  if (ty == 's' || (ty == 'c' && memchr(v, ',', n) == nullptr)) {
    out->append(v, n);
    return;
  }
  bool first = true;
  size_t a = 0;
  while (a <= n) {
    size_t e = a;
    while (e < n && v[e] != ',') ++e;
    const bool dot = e - a == 1 && v[a] == '.';
    if (!(dot && ty == 'c')) {
      if (!first) out->push_back(',');
      first = false;
      if (dot) out->push_back('.');
      else if (ty == 'i') print_i32(v + a, e - a, out);
      else if (ty == 'f') rust_f32_display(parse_f32(v + a, e - a), out);
      else out->append(v + a, e - a);
    }
    a = e + 1;
  }
}
// GT: alleles split on '/' and '|'; allele i (i >= 1) carries the separator in front of it; allele 0's phasing before VCF 4.4
// is "unphased when any separator is '/'" (noodles-vcf record::samples::series::value::genotype, as published); the reference
// prints the separator in front of allele i from allele i - 1's phasing (lazy_array_builder.rs:331-360)
inline void print_genotype(const char* v, size_t n, std::string* out) {
  std::vector<std::pair<std::string, char>> alleles;  // (text, phasing '/' or '|')
  const bool any_unphased = memchr(v, '/', n) != nullptr;
  size_t a = 0;
  char sep = 0;
  for (size_t i = 0; i <= n; ++i) {
    if (i == n || v[i] == '/' || v[i] == '|') {
      std::string al(v + a, i - a);
      if (al.empty()) throw std::runtime_error("invalid genotype '" + std::string(v, n) + "'");
      if (al != ".") {  // an allele index prints through usize's Display
        for (char ch : al)
          if (ch < '0' || ch > '9') throw std::runtime_error("invalid genotype '" + std::string(v, n) + "'");
        size_t z = 0;
        while (z + 1 < al.size() && al[z] == '0') ++z;
        al = al.substr(z);
      }
      alleles.emplace_back(al, alleles.empty() ? (any_unphased ? '/' : '|') : sep);
      if (i < n) sep = v[i];
      a = i + 1;
    }
  }
  for (size_t i = 0; i < alleles.size(); ++i) {
    if (i) out->push_back(alleles[i - 1].second);
    out->append(alleles[i].first);
  }
}
}  // namespace vcf_text_detail

// the `info` column of one record (field 8 of the line)
template <typename ParseF32>
inline void vcf_info_string(const char* p, size_t n, const VcfKeyTypes& types, ParseF32 parse_f32, std::string* out) {
  out->clear();
  if (n == 0 || (n == 1 && p[0] == '.')) return;
  size_t i = 0;
  bool first = true;
  while (i <= n) {
    size_t j = i;
    while (j < n && p[j] != ';') ++j;
    if (j > i) {
      size_t eq = i;
      while (eq < j && p[eq] != '=') ++eq;
      const char ty = types.info_type(p + i, eq - i);
      if (!first) out->push_back(';');
      first = false;
      out->append(p + i, eq - i);
      out->push_back('=');
      if (ty == 'b') {
        out->append("true");
      } else {
        const char* v = eq < j ? p + eq + 1 : p + j;
        const size_t vl = eq < j ? j - eq - 1 : 0;
        if (eq >= j || vl == 0 || (vl == 1 && v[0] == '.'))
          throw std::runtime_error("INFO key '" + std::string(p + i, eq - i) + "' has no value: the reference's info-as-string builder panics here (lazy_array_builder.rs:223)");
        vcf_text_detail::print_value(ty, v, vl, parse_f32, out);
      }
    }
    i = j + 1;
  }
}

// the `formats` column of one record: `format` = field 9, `samples` = everything behind it (TAB-separated), both may be absent
template <typename ParseF32>
inline void vcf_formats_string(const char* format, size_t nf, const char* samples, size_t ns, const VcfKeyTypes& types, ParseF32 parse_f32,
                               std::string* out) {
  out->clear();
  std::vector<char> tys;
  if (nf && !(nf == 1 && format[0] == '.')) {
    size_t a = 0;
    for (size_t i = 0; i <= nf; ++i)
      if (i == nf || format[i] == ':') {
        tys.push_back(types.format_type(format + a, i - a));
        a = i + 1;
      }
    out->append(format, nf);
  }
  out->push_back('\t');
  if (tys.empty()) return;
  size_t s0 = 0;
  bool first_sample = true;
  for (size_t i = 0; i <= ns; ++i) {
    if (i != ns && samples[i] != '\t') continue;
    if (!first_sample) out->push_back('\t');
    first_sample = false;
    size_t a = s0, k = 0;
    bool first_value = true;
    for (size_t q = s0; q <= i; ++q) {
      if (q != i && samples[q] != ':') continue;
      if (k < tys.size()) {
        const char* v = samples + a;
        const size_t vl = q - a;
        if (vl == 0 || (vl == 1 && v[0] == '.'))
          throw std::runtime_error("a sample value is missing: the reference's formats-as-string builder panics here (lazy_array_builder.rs:326)");
        if (!first_value) out->push_back(':');
        first_value = false;
        if (tys[k] == 'g') vcf_text_detail::print_genotype(v, vl, out);
        else vcf_text_detail::print_value(tys[k], v, vl, parse_f32, out);
      }
      ++k;
      a = q + 1;
    }
    s0 = i + 1;
  }
}

}  // namespace exon
