// cli.cpp -- exon-hip-cli: the build's own small command line.
//
// Mirrors the part of exon-cli that the hot path needs (exon/exon-cli/src/main.rs:31-146: `-c <sql>...`,
// `-f <file>...`, `-q`, print a table): a session with external tables, the *_scan table functions and the
// four fused query shapes.  Plain `SELECT COUNT(*)` without a predicate is CPU plumbing (decode + count, no
// GPU, like config 1); predicates / GROUP BY run through exon_hip_plan_* / exon_hip_stream_* on the GPU.
//
//   SET exon.vcf_parse_info = true;
//   CREATE EXTERNAL TABLE t STORED AS FASTA|FASTQ|VCF|BAM|INDEXED_VCF|INDEXED_BAM [OPTIONS (compression gzip)] LOCATION '<path|dir>';
//   SELECT COUNT(*) FROM t | fasta_scan('<p>'[, 'gzip']) | fastq_scan(..) | vcf_scan(..) | bam_scan(..)
//                         | vcf_indexed_scan('<p>', '<region>') | bam_indexed_scan('<p>', '<region>')
//   SELECT COUNT(*) FROM v WHERE chrom = '7' AND pos >= 50000000 AND pos <= 100000000            -- K2
//   SELECT COUNT(*) FROM v WHERE vcf_region_filter('7:50000000-100000000', chrom[, pos]) [= true] -- pushed down
//   SELECT COUNT(*) FROM b WHERE bam_region_filter('chr1:1-100', reference, start, end) [= true]  -- pushed down
//   SELECT reference, COUNT(*) FROM b WHERE flag & 1284 = 0 AND CAST(mapping_quality AS INT) >= 30 GROUP BY reference  -- K3
//   SELECT filter, AVG(qual), COUNT(*) FROM v WHERE info."AF" > 0.01 GROUP BY filter            -- K4
//   SELECT * FROM fastq_quality_histogram('<p>'[, 'gzip'])                                        -- K5
//   SELECT COUNT(*) FROM <bam table> WHERE bam_region_filter('<r>', reference, start, end)          -- K6 (plain table; INDEXED_BAM plans chunks on the host)
//   DROP TABLE t;
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/exon_hip.h"

namespace {

struct Err : std::runtime_error {
  using std::runtime_error::runtime_error;
};

// ---- tokens -------------------------------------------------------------------------------------------
struct Tok {
  enum Kind { Ident, QIdent, Str, Num, Sym, End } kind;
  std::string text;
};

std::vector<Tok> tokenize(const std::string& s) {
  std::vector<Tok> out;
  size_t i = 0;
  while (i < s.size()) {
    const char c = s[i];
    if (isspace((unsigned char)c)) { ++i; continue; }
    if (c == '-' && i + 1 < s.size() && s[i + 1] == '-') { while (i < s.size() && s[i] != '\n') ++i; continue; }
    if (isalpha((unsigned char)c) || c == '_') {
      size_t j = i;
      while (j < s.size() && (isalnum((unsigned char)s[j]) || s[j] == '_' || s[j] == '.')) ++j;
      out.push_back({Tok::Ident, s.substr(i, j - i)});
      i = j;
    } else if (isdigit((unsigned char)c)) {
      size_t j = i;
      while (j < s.size() && (isalnum((unsigned char)s[j]) || s[j] == '.' || ((s[j] == '-' || s[j] == '+') && (s[j - 1] == 'e' || s[j - 1] == 'E')))) ++j;
      out.push_back({Tok::Num, s.substr(i, j - i)});
      i = j;
    } else if (c == '\'' || c == '"') {
      std::string v;
      size_t j = i + 1;
      for (;; ++j) {
        if (j >= s.size()) throw Err("unterminated string literal");
        if (s[j] == c) {
          if (j + 1 < s.size() && s[j + 1] == c) { v += c; ++j; continue; }
          break;
        }
        v += s[j];
      }
      out.push_back({c == '\'' ? Tok::Str : Tok::QIdent, v});
      i = j + 1;
    } else {
      std::string sym(1, c);
      if ((c == '<' || c == '>' || c == '!') && i + 1 < s.size() && (s[i + 1] == '=' || (c == '<' && s[i + 1] == '>'))) sym += s[i + 1];
      out.push_back({Tok::Sym, sym});
      i += sym.size();
    }
  }
  out.push_back({Tok::End, ""});
  return out;
}

std::string lower(std::string s) {
  for (auto& c : s) c = (char)tolower((unsigned char)c);
  return s;
}

struct Parser {
  std::vector<Tok> t;
  size_t p = 0;
  const Tok& peek(size_t k = 0) const { return t[std::min(p + k, t.size() - 1)]; }
  bool is_kw(const char* kw, size_t k = 0) const { return peek(k).kind == Tok::Ident && lower(peek(k).text) == kw; }
  bool is_sym(const char* s, size_t k = 0) const { return peek(k).kind == Tok::Sym && peek(k).text == s; }
  bool accept_kw(const char* kw) { if (is_kw(kw)) { ++p; return true; } return false; }
  bool accept_sym(const char* s) { if (is_sym(s)) { ++p; return true; } return false; }
  void expect_kw(const char* kw) { if (!accept_kw(kw)) throw Err(std::string("expected ") + kw + " near '" + peek().text + "'"); }
  void expect_sym(const char* s) { if (!accept_sym(s)) throw Err(std::string("expected '") + s + "' near '" + peek().text + "'"); }
  std::string ident() {
    if (peek().kind != Tok::Ident && peek().kind != Tok::QIdent) throw Err("expected identifier near '" + peek().text + "'");
    return t[p++].text;
  }
  std::string str() {
    if (peek().kind != Tok::Str) throw Err("expected string literal near '" + peek().text + "'");
    return t[p++].text;
  }
  std::string number() {
    std::string sign;
    if (accept_sym("-")) sign = "-";
    if (peek().kind != Tok::Num) throw Err("expected number near '" + peek().text + "'");
    return sign + t[p++].text;
  }
  bool at_end() const { return peek().kind == Tok::End; }
};

// ---- session --------------------------------------------------------------------------------------------
struct Table {
  int format = 0;
  bool indexed = false;
  int compression = EXON_HIP_COMPRESSION_AUTO;
  std::string location, extension;
};

struct Source {  // resolved FROM clause
  int format = 0;
  int compression = EXON_HIP_COMPRESSION_AUTO;
  std::vector<std::string> files;
  std::string region;  // from *_indexed_scan
  bool indexed = false;
};

struct Session {
  std::map<std::string, Table> tables;
  bool vcf_parse_info = false;
  bool quiet = false;
  exon_hip_ctx* ctx = nullptr;
  ~Session() { if (ctx) exon_hip_ctx_destroy(ctx); }
  exon_hip_ctx* gpu() {
    if (!ctx && exon_hip_ctx_create(0, &ctx) != 0) throw Err(std::string("GPU required for this query: ") + exon_hip_last_error(nullptr));
    return ctx;
  }
};

void ck(exon_hip_ctx* ctx, int rc) {
  if (rc < 0) {
    std::string m = ctx ? exon_hip_last_error(ctx) : "";
    if (m.empty()) m = exon_hip_last_error(nullptr);
    throw Err(m);
  }
}

bool ends_with(const std::string& s, const std::string& suf) { return s.size() >= suf.size() && s.compare(s.size() - suf.size(), suf.size(), suf) == 0; }

// object_store_files_from_table_path (exon-common/src/object_store_files_from_table_path.rs:21-47): a file, or every
// file under a directory whose name carries the format's extension (optionally + compression suffix)
void list_files(const std::string& path, const std::vector<std::string>& exts, std::vector<std::string>* out) {
  struct stat st;
  if (stat(path.c_str(), &st) != 0) throw Err("no such file or directory: " + path);
  if (!S_ISDIR(st.st_mode)) { out->push_back(path); return; }
  DIR* d = opendir(path.c_str());
  if (!d) throw Err("cannot list " + path);
  std::vector<std::string> names;
  while (dirent* e = readdir(d)) if (e->d_name[0] != '.') names.push_back(e->d_name);
  closedir(d);
  std::sort(names.begin(), names.end());
  for (const auto& n : names) {
    const std::string full = path + (ends_with(path, "/") ? "" : "/") + n;
    if (stat(full.c_str(), &st) != 0) continue;
    if (S_ISDIR(st.st_mode)) { list_files(full, exts, out); continue; }
    for (const auto& e : exts)
      if (ends_with(n, e) || ends_with(n, e + ".gz") || ends_with(n, e + ".bgz")) { out->push_back(full); break; }
  }
}

std::vector<std::string> format_exts(int format, const std::string& custom) {
  if (!custom.empty()) return {"." + custom};
  switch (format) {
    case EXON_HIP_FORMAT_FASTA: return {".fasta", ".fa", ".fna", ".faa"};
    case EXON_HIP_FORMAT_FASTQ: return {".fastq", ".fq"};
    case EXON_HIP_FORMAT_VCF: return {".vcf"};
    case EXON_HIP_FORMAT_SAM: return {".sam"};
    case EXON_HIP_FORMAT_BCF: return {".bcf"};
    case EXON_HIP_FORMAT_CRAM: return {".cram"};
    default: return {".bam"};
  }
}

int format_of(const std::string& name, bool* indexed) {
  std::string f = lower(name);
  *indexed = false;
  if (f.rfind("indexed_", 0) == 0) { *indexed = true; f = f.substr(8); }
  if (f == "fasta") return EXON_HIP_FORMAT_FASTA;
  if (f == "fastq") return EXON_HIP_FORMAT_FASTQ;
  if (f == "vcf") return EXON_HIP_FORMAT_VCF;
  if (f == "bam") return EXON_HIP_FORMAT_BAM;
  if (f == "sam") return EXON_HIP_FORMAT_SAM;
  if (f == "bcf") return EXON_HIP_FORMAT_BCF;
  if (f == "cram") return EXON_HIP_FORMAT_CRAM;
  throw Err("unsupported file type " + name);
}

int compression_of(const std::string& v) {
  const std::string c = lower(v);
  if (c == "gzip" || c == "gz" || c == "bgzip") return EXON_HIP_COMPRESSION_GZIP;
  if (c == "none" || c == "uncompressed" || c.empty()) return EXON_HIP_COMPRESSION_NONE;
  throw Err("unsupported compression " + v);
}

// ---- output ---------------------------------------------------------------------------------------------
void print_table(const std::vector<std::string>& cols, const std::vector<std::vector<std::string>>& rows, bool quiet) {
  std::vector<size_t> w(cols.size());
  for (size_t i = 0; i < cols.size(); ++i) w[i] = cols[i].size();
  for (const auto& r : rows) for (size_t i = 0; i < cols.size(); ++i) w[i] = std::max(w[i], r[i].size());
  auto line = [&]() { for (size_t i = 0; i < cols.size(); ++i) printf("+%s", std::string(w[i] + 2, '-').c_str()); printf("+\n"); };
  line();
  for (size_t i = 0; i < cols.size(); ++i) printf("| %-*s ", (int)w[i], cols[i].c_str());
  printf("|\n");
  line();
  for (const auto& r : rows) {
    for (size_t i = 0; i < cols.size(); ++i) printf("| %-*s ", (int)w[i], r[i].c_str());
    printf("|\n");
  }
  line();
  if (!quiet) printf("%zu row(s) fetched.\n", rows.size());
}

std::string fmt_f64(double v) {
  char b[64];
  snprintf(b, sizeof b, "%.17g", v);
  // shortest representation that round-trips, like Rust's Display for f64
  for (int p = 1; p < 17; ++p) {
    char t[64];
    snprintf(t, sizeof t, "%.*g", p, v);
    if (strtod(t, nullptr) == v) { snprintf(b, sizeof b, "%s", t); break; }
  }
  std::string s = b;
  if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
  return s;
}

// ---- execution --------------------------------------------------------------------------------------------
struct ScanGuard {
  exon_hip_scan* s = nullptr;
  ~ScanGuard() { if (s) exon_hip_scan_close(s); }
};
struct StreamGuard {
  exon_hip_plan* p = nullptr;
  exon_hip_stream* s = nullptr;
  ~StreamGuard() { if (s) exon_hip_stream_close(s); if (p) exon_hip_plan_destroy(p); }
};

// VCF / FASTQ queries that end in a fused GPU kernel ship the text to HBM and parse it there (EXON_HIP_GPU_PARSE=0: host decode)
bool gpu_parse_enabled() {
  const char* v = getenv("EXON_HIP_GPU_PARSE");
  return !(v && v[0] == '0');
}

void open_scan(const Source& src, const std::string& file, const char* info_field, const std::string& region, ScanGuard* g,
               bool for_gpu_query = false) {
  exon_hip_scan_options o;
  memset(&o, 0, sizeof o);
  o.format = src.format;
  o.compression = src.compression;
  o.info_field = info_field;
  o.region = region.empty() ? nullptr : region.c_str();
  // INDEXED_* tables / *_indexed_scan: plan BGZF chunks from <file>.tbi / <file>.bai
  // (exon-core/src/datasources/indexed_file/indexed_bgzf_file.rs:129-155)
  o.use_index = (src.indexed && !region.empty()) ? 1 : 0;
  o.gpu_parse = (for_gpu_query && (src.format == EXON_HIP_FORMAT_VCF || src.format == EXON_HIP_FORMAT_FASTQ || src.format == EXON_HIP_FORMAT_BAM || src.format == EXON_HIP_FORMAT_BCF || src.format == EXON_HIP_FORMAT_SAM) &&
                 region.empty() && gpu_parse_enabled()) ? 1 : 0;
  ck(nullptr, exon_hip_scan_open(file.c_str(), &o, &g->s));
}

// CPU plumbing: decode and count rows (optionally with a pushed-down region filter)
int64_t count_rows(const Source& src, const std::string& region) {
  int64_t total = 0;
  for (const auto& f : src.files) {
    ScanGuard g;
    open_scan(src, f, nullptr, region, &g);
    for (;;) {
      struct ArrowArray b;
      const int rc = exon_hip_scan_next(g.s, &b);
      if (rc == 1) break;
      ck(nullptr, rc);
      total += b.length;
      b.release(&b);
    }
  }
  return total;
}

struct Predicate {  // what the WHERE clause turned into
  enum Kind { None, Region, PushedRegion, FlagMapq, InfoCmp } kind = None;
  std::string chrom; int64_t start = 1, end = INT64_MAX;      // Region
  std::string region;                                          // PushedRegion
  int32_t flag_mask = 0, flag_value = 0, mapq_min = INT32_MIN; // FlagMapq
  std::string info_field; int cmp_op = 0; double literal = 0;  // InfoCmp
};

int cmp_code(const std::string& s) {
  if (s == ">") return EXON_HIP_GT;
  if (s == ">=") return EXON_HIP_GE;
  if (s == "<") return EXON_HIP_LT;
  if (s == "<=") return EXON_HIP_LE;
  if (s == "=") return EXON_HIP_EQ;
  if (s == "!=" || s == "<>") return EXON_HIP_NE;
  throw Err("unsupported comparison " + s);
}

Predicate parse_where(Parser& ps, int format) {
  Predicate pr;
  // pushed-down marker UDFs (exon-core/src/udfs/vcf/vcf_region_filter.rs:23-75, udfs/sam/bam_region_filter.rs:23-86)
  if (ps.is_kw("vcf_region_filter") || ps.is_kw("bam_region_filter")) {
    ps.ident();
    ps.expect_sym("(");
    pr.kind = Predicate::PushedRegion;
    pr.region = ps.str();
    while (ps.accept_sym(",")) ps.ident();
    ps.expect_sym(")");
    if (ps.accept_sym("=")) ps.expect_kw("true");
    return pr;
  }
  if (ps.is_kw("region_match")) {  // region_match(chrom, pos, 'r')
    ps.ident(); ps.expect_sym("("); ps.ident(); ps.expect_sym(","); ps.ident(); ps.expect_sym(",");
    char name[512];
    ck(nullptr, exon_hip_parse_region(ps.str().c_str(), name, sizeof name, &pr.start, &pr.end));
    pr.chrom = name;
    pr.kind = Predicate::Region;
    ps.expect_sym(")");
    return pr;
  }
  bool first = true;
  while (first || ps.accept_kw("and")) {
    first = false;
    if (ps.is_kw("cast")) {  // CAST(mapping_quality AS INT) >= q
      ps.ident(); ps.expect_sym("("); ps.ident(); ps.expect_kw("as"); ps.ident(); ps.expect_sym(")");
      const std::string op = ps.peek().text; ps.p++;
      const int q = atoi(ps.number().c_str());
      if (op == ">=") pr.mapq_min = q; else if (op == ">") pr.mapq_min = q + 1; else throw Err("only >= / > on CAST(mapping_quality AS INT)");
      pr.kind = Predicate::FlagMapq;
      continue;
    }
    const std::string col = lower(ps.ident());
    if (col == "chrom") { ps.expect_sym("="); pr.chrom = ps.str(); pr.kind = Predicate::Region; }
    else if (col == "pos") {
      pr.kind = Predicate::Region;
      if (ps.accept_kw("between")) { pr.start = atoll(ps.number().c_str()); ps.expect_kw("and"); pr.end = atoll(ps.number().c_str()); }
      else {
        const std::string op = ps.peek().text; ps.p++;
        const int64_t v = atoll(ps.number().c_str());
        if (op == ">=") pr.start = std::max(pr.start, v); else if (op == ">") pr.start = std::max(pr.start, v + 1);
        else if (op == "<=") pr.end = std::min(pr.end, v); else if (op == "<") pr.end = std::min(pr.end, v - 1);
        else if (op == "=") { pr.start = std::max(pr.start, v); pr.end = std::min(pr.end, v); }
        else throw Err("unsupported operator on pos: " + op);
      }
    } else if (col == "flag") {  // flag & M = V
      ps.expect_sym("&"); pr.flag_mask = atoi(ps.number().c_str()); ps.expect_sym("="); pr.flag_value = atoi(ps.number().c_str());
      pr.kind = Predicate::FlagMapq;
    } else if (col == "info" || col.rfind("info.", 0) == 0) {  // info."AF" <op> lit
      if (col.size() > 5) pr.info_field = col.substr(5);  // unquoted identifiers fold to lower case, as in DataFusion
      else { ps.accept_sym("."); pr.info_field = ps.ident(); }
      pr.cmp_op = cmp_code(ps.peek().text); ps.p++;
      pr.literal = atof(ps.number().c_str());
      pr.kind = Predicate::InfoCmp;
    } else {
      throw Err("unsupported predicate column '" + col + "' for this build (the fused shapes are listed in --help)");
    }
  }
  (void)format;
  return pr;
}

Source resolve_from(Session& se, Parser& ps) {
  Source src;
  const std::string name = ps.ident();
  const std::string lname = lower(name);
  if (ps.accept_sym("(")) {  // table function: <fmt>_scan / <fmt>_indexed_scan / fastq_quality_histogram
    const std::string path = ps.str();
    std::string arg2;
    if (ps.accept_sym(",")) arg2 = ps.str();
    ps.expect_sym(")");
    const size_t us = lname.find('_');
    bool idx = false;
    src.format = format_of(lname.substr(0, us), &idx);
    src.indexed = lname.find("_indexed_scan") != std::string::npos;
    if (src.indexed) src.region = arg2;
    else if (!arg2.empty()) src.compression = compression_of(arg2);
    list_files(path, format_exts(src.format, ""), &src.files);
    return src;
  }
  auto it = se.tables.find(lname);
  if (it == se.tables.end()) throw Err("table '" + name + "' not found");
  src.format = it->second.format;
  src.compression = it->second.compression;
  src.indexed = it->second.indexed;
  list_files(it->second.location, format_exts(src.format, it->second.extension), &src.files);
  return src;
}

void exec_select(Session& se, Parser& ps) {
  // projection
  std::vector<std::string> proj;
  bool star = false;
  do {
    if (ps.accept_sym("*")) { star = true; continue; }
    std::string item = lower(ps.ident());
    if (ps.accept_sym("(")) {
      if (ps.accept_sym("*")) item += "(*)"; else item += "(" + lower(ps.ident()) + ")";
      ps.expect_sym(")");
    }
    if (ps.accept_kw("as")) ps.ident();
    else if (ps.peek().kind == Tok::Ident && !ps.is_kw("from")) ps.ident();  // bare alias
    proj.push_back(item);
  } while (ps.accept_sym(","));
  ps.expect_kw("from");
  const bool hist = ps.is_kw("fastq_quality_histogram");
  Source src = resolve_from(se, ps);
  Predicate pr;
  if (ps.accept_kw("where")) pr = parse_where(ps, src.format);
  std::string group_by;
  if (ps.accept_kw("group")) { ps.expect_kw("by"); group_by = lower(ps.ident()); while (ps.accept_sym(",")) ps.ident(); }
  if (ps.accept_kw("order")) { ps.expect_kw("by"); while (!ps.at_end() && !ps.is_sym(";")) ps.p++; }
  if (!ps.at_end()) throw Err("unexpected '" + ps.peek().text + "'");

  if (src.indexed && src.region.empty() && pr.kind != Predicate::PushedRegion)
    throw Err("an INDEXED table requires a region filter");  // slt/vcf-indexed-tests.slt:6-8, :48-49
  if (!src.region.empty()) { pr.kind = Predicate::PushedRegion; pr.region = src.region; }

  if (hist) {  // K5
    exon_hip_ctx* ctx = se.gpu();
    const int lmax = 512;
    std::vector<int64_t> total((size_t)lmax * 256, 0);
    for (const auto& f : src.files) {
      ScanGuard g; open_scan(src, f, nullptr, "", &g, true);
      StreamGuard sg;
      exon_hip_plan_desc d; memset(&d, 0, sizeof d);
      d.kind = EXON_HIP_PLAN_QUAL_POS_HIST; d.lmax = lmax; d.columns[0] = 3;
      ck(ctx, exon_hip_plan_create(ctx, &d, &sg.p));
      ck(ctx, exon_hip_stream_open(sg.p, 0, &sg.s));
      ck(ctx, exon_hip_stream_consume_scan(sg.s, g.s, nullptr));
      std::vector<int64_t> h((size_t)lmax * 256);
      ck(ctx, exon_hip_stream_finish(sg.s, h.data(), nullptr));
      for (size_t i = 0; i < h.size(); ++i) total[i] += h[i];
    }
    std::vector<std::vector<std::string>> rows;
    for (int p = 0; p < lmax; ++p)
      for (int b = 0; b < 256; ++b)
        if (total[(size_t)p * 256 + b]) rows.push_back({std::to_string(p + 1), std::to_string(b - 33), std::to_string(total[(size_t)p * 256 + b])});
    print_table({"position", "quality_score", "count(*)"}, rows, se.quiet);
    return;
  }

  const bool count_only = proj.size() == 1 && proj[0] == "count(*)" && group_by.empty() && !star;
  if (count_only && pr.kind == Predicate::PushedRegion && !src.indexed && gpu_parse_enabled() &&
      (src.format == EXON_HIP_FORMAT_BAM || src.format == EXON_HIP_FORMAT_SAM || src.format == EXON_HIP_FORMAT_CRAM)) {  // K6: interval overlap on the GPU
    exon_hip_ctx* ctx = se.gpu();
    char name[512];
    int64_t a = 1, b = INT64_MAX;
    ck(nullptr, exon_hip_parse_region(pr.region.c_str(), name, sizeof name, &a, &b));
    int64_t total = 0;
    for (const auto& f : src.files) {
      ScanGuard g; open_scan(src, f, nullptr, "", &g, true);
      int32_t rid = -1;
      ck(nullptr, exon_hip_scan_dictionary_intern(g.s, 2, name, &rid));
      if (rid < 0) continue;  // the reference is not in this file's header
      StreamGuard sg;
      exon_hip_plan_desc d; memset(&d, 0, sizeof d);
      d.kind = EXON_HIP_PLAN_OVERLAP_COUNT; d.region_chrom_id = rid; d.region_start = a; d.region_end = b;
      d.columns[0] = 2; d.columns[1] = 3; d.columns[2] = 4;
      ck(ctx, exon_hip_plan_create(ctx, &d, &sg.p));
      ck(ctx, exon_hip_stream_open(sg.p, 0, &sg.s));
      ck(ctx, exon_hip_stream_consume_scan(sg.s, g.s, nullptr));
      int64_t c = 0;
      ck(ctx, exon_hip_stream_finish(sg.s, &c, nullptr));
      total += c;
    }
    print_table({"count(*)"}, {{std::to_string(total)}}, se.quiet);
    return;
  }
  if (count_only && (pr.kind == Predicate::None || pr.kind == Predicate::PushedRegion)) {
    // config-1 plumbing / pushed-down region: the decoder's per-record filter is the only filter
    print_table({"count(*)"}, {{std::to_string(count_rows(src, pr.region))}}, se.quiet);
    return;
  }
  if (count_only && pr.kind == Predicate::Region && (src.format == EXON_HIP_FORMAT_VCF || src.format == EXON_HIP_FORMAT_BCF)) {  // K2
    exon_hip_ctx* ctx = se.gpu();
    int64_t total = 0;
    for (const auto& f : src.files) {
      ScanGuard g; open_scan(src, f, nullptr, "", &g, true);
      int32_t cid = -1;
      ck(nullptr, exon_hip_scan_dictionary_intern(g.s, 0, pr.chrom.c_str(), &cid));
      StreamGuard sg;
      exon_hip_plan_desc d; memset(&d, 0, sizeof d);
      d.kind = EXON_HIP_PLAN_REGION_COUNT; d.region_chrom_id = cid; d.region_start = pr.start; d.region_end = pr.end;
      d.columns[0] = 0; d.columns[1] = 1;
      if (pr.end < pr.start) continue;  // empty interval
      ck(ctx, exon_hip_plan_create(ctx, &d, &sg.p));
      ck(ctx, exon_hip_stream_open(sg.p, 0, &sg.s));
      ck(ctx, exon_hip_stream_consume_scan(sg.s, g.s, nullptr));
      int64_t c = 0;
      ck(ctx, exon_hip_stream_finish(sg.s, &c, nullptr));
      total += c;
    }
    print_table({"count(*)"}, {{std::to_string(total)}}, se.quiet);
    return;
  }
  if (pr.kind == Predicate::FlagMapq && (src.format == EXON_HIP_FORMAT_BAM || src.format == EXON_HIP_FORMAT_SAM || src.format == EXON_HIP_FORMAT_CRAM) && group_by == "reference") {  // K3
    exon_hip_ctx* ctx = se.gpu();
    std::map<std::string, int64_t> merged;  // AggregateExec(Final): merge per-file partials by key
    int64_t null_group = 0;
    std::vector<std::string> order;
    for (const auto& f : src.files) {
      ScanGuard g; open_scan(src, f, nullptr, "", &g, true);
      int32_t R = 0;
      ck(nullptr, exon_hip_scan_dictionary_size(g.s, 2, &R));
      StreamGuard sg;
      exon_hip_plan_desc d; memset(&d, 0, sizeof d);
      d.kind = EXON_HIP_PLAN_FLAG_MAPQ_GROUP_COUNT; d.n_groups = R; d.flag_mask = pr.flag_mask; d.flag_value = pr.flag_value;
      d.mapq_min = pr.mapq_min; d.columns[0] = 0; d.columns[1] = 1; d.columns[2] = 2;
      ck(ctx, exon_hip_plan_create(ctx, &d, &sg.p));
      ck(ctx, exon_hip_stream_open(sg.p, 0, &sg.s));
      ck(ctx, exon_hip_stream_consume_scan(sg.s, g.s, nullptr));
      std::vector<int64_t> c((size_t)R + 1);
      ck(ctx, exon_hip_stream_finish(sg.s, c.data(), nullptr));
      for (int32_t r = 0; r < R; ++r)
        if (c[(size_t)r]) {
          const char* nm; ck(nullptr, exon_hip_scan_dictionary_value(g.s, 2, r, &nm));
          if (!merged.count(nm)) order.push_back(nm);
          merged[nm] += c[(size_t)r];
        }
      null_group += c[(size_t)R];
    }
    std::vector<std::vector<std::string>> rows;
    for (const auto& k : order) rows.push_back({k, std::to_string(merged[k])});
    if (null_group) rows.push_back({"NULL", std::to_string(null_group)});
    print_table({"reference", "count(*)"}, rows, se.quiet);
    return;
  }
  if (pr.kind == Predicate::InfoCmp && (src.format == EXON_HIP_FORMAT_VCF || src.format == EXON_HIP_FORMAT_BCF) && group_by == "filter") {  // K4
    if (!se.vcf_parse_info) throw Err("info." + pr.info_field + " needs `SET exon.vcf_parse_info = true` (info is a Utf8 column otherwise)");
    exon_hip_ctx* ctx = se.gpu();
    struct Acc { double sum = 0; int64_t cnt = 0, rows = 0; };
    std::map<std::string, Acc> merged;
    std::vector<std::string> order;
    for (const auto& f : src.files) {
      ScanGuard g; open_scan(src, f, pr.info_field.c_str(), "", &g, true);
      StreamGuard sg;
      const int G = 256;  // distinct FILTER lists supported per file (8 in registers + LDS overflow table)
      exon_hip_plan_desc d; memset(&d, 0, sizeof d);
      d.kind = EXON_HIP_PLAN_CMP_AVG_BY_GROUP; d.n_groups = G; d.cmp_op = pr.cmp_op; d.threshold = pr.literal;
      d.columns[0] = 4; d.columns[1] = 2; d.columns[2] = 3;
      ck(ctx, exon_hip_plan_create(ctx, &d, &sg.p));
      ck(ctx, exon_hip_stream_open(sg.p, 0, &sg.s));
      ck(ctx, exon_hip_stream_consume_scan(sg.s, g.s, nullptr));
      int32_t nd = 0;
      ck(nullptr, exon_hip_scan_dictionary_size(g.s, 3, &nd));
      if (nd > G) throw Err("more than " + std::to_string(G) + " distinct FILTER lists in one file");
      std::vector<int64_t> c((size_t)2 * G);
      std::vector<double> s((size_t)G);
      ck(ctx, exon_hip_stream_finish(sg.s, c.data(), s.data()));
      for (int32_t k = 0; k < nd; ++k)
        if (c[(size_t)(G + k)]) {
          const char* nm; ck(nullptr, exon_hip_scan_dictionary_value(g.s, 3, k, &nm));
          if (!merged.count(nm)) order.push_back(nm);
          Acc& a = merged[nm];
          a.sum += s[(size_t)k]; a.cnt += c[(size_t)k]; a.rows += c[(size_t)(G + k)];
        }
    }
    std::vector<std::vector<std::string>> rows;
    for (const auto& k : order) {
      const Acc& a = merged[k];
      std::string list = "[";  // List<Utf8> rendered like datafusion: [a, b]
      size_t st = 0;
      while (!k.empty() && st <= k.size()) {
        const size_t sc = k.find(';', st);
        list += (st ? ", " : "") + k.substr(st, sc == std::string::npos ? std::string::npos : sc - st);
        if (sc == std::string::npos) break;
        st = sc + 1;
      }
      list += "]";
      rows.push_back({list, a.cnt ? fmt_f64(a.sum / (double)a.cnt) : "NULL", std::to_string(a.rows)});
    }
    print_table({"filter", "avg(qual)", "count(*)"}, rows, se.quiet);
    return;
  }
  throw Err("query shape not supported by this build: supported shapes are listed in `exon-hip-cli --help`");
}

void exec_statement(Session& se, const std::string& sql) {
  Parser ps{tokenize(sql)};
  if (ps.at_end()) return;
  if (ps.accept_kw("set")) {
    const std::string key = lower(ps.ident());
    if (!ps.accept_sym("=")) ps.accept_kw("to");
    const std::string val = ps.peek().kind == Tok::Str ? ps.str() : lower(ps.ident());
    if (key == "exon.vcf_parse_info") se.vcf_parse_info = (val == "true");
    return;  // other exon.* options (config/mod.rs:65-78) are accepted and ignored
  }
  if (ps.accept_kw("create")) {
    ps.expect_kw("external"); ps.expect_kw("table");
    const std::string name = lower(ps.ident());
    Table t;
    bool have_loc = false;
    while (!ps.at_end()) {
      if (ps.accept_kw("stored")) { ps.expect_kw("as"); t.format = format_of(ps.ident(), &t.indexed); }
      else if (ps.accept_kw("location")) { t.location = ps.str(); have_loc = true; }
      else if (ps.accept_kw("partitioned")) { ps.expect_kw("by"); ps.expect_sym("("); while (!ps.accept_sym(")")) ps.p++; }
      else if (ps.accept_kw("options")) {
        ps.expect_sym("(");
        while (!ps.accept_sym(")")) {
          std::string k = ps.peek().kind == Tok::Str ? ps.str() : ps.ident();
          std::string v = ps.peek().kind == Tok::Str ? ps.str() : ps.ident();
          k = lower(k);
          if (k == "compression" || ends_with(k, ".compression")) t.compression = compression_of(v);
          if (k == "file_extension" || ends_with(k, ".file_extension")) t.extension = v;
          if (k == "indexed" && lower(v) == "true") t.indexed = true;
          ps.accept_sym(",");
        }
      } else throw Err("unexpected '" + ps.peek().text + "' in CREATE EXTERNAL TABLE");
    }
    if (!t.format || !have_loc) throw Err("CREATE EXTERNAL TABLE needs STORED AS and LOCATION");
    se.tables[name] = t;
    return;
  }
  if (ps.accept_kw("drop")) { ps.expect_kw("table"); se.tables.erase(lower(ps.ident())); return; }
  if (ps.accept_kw("select")) { exec_select(se, ps); return; }
  throw Err("unsupported statement starting with '" + ps.peek().text + "'");
}

void exec_script(Session& se, const std::string& text) {
  std::string cur;
  bool inq = false;
  for (char c : text) {
    if (c == '\'') inq = !inq;
    if (c == ';' && !inq) { exec_statement(se, cur); cur.clear(); }
    else cur += c;
  }
  exec_statement(se, cur);
}

}  // namespace

int main(int argc, char** argv) {
  Session se;
  std::vector<std::string> commands, files;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-c" || a == "--command") { while (i + 1 < argc && argv[i + 1][0] != '-') commands.push_back(argv[++i]); }
    else if (a == "-f" || a == "--file") { while (i + 1 < argc && argv[i + 1][0] != '-') files.push_back(argv[++i]); }
    else if (a == "-q" || a == "--quiet") se.quiet = true;
    else if (a == "-h" || a == "--help") {
      printf("exon-hip-cli [-q] -c '<sql>'... | -f <file>...\n"
             "  tables:    CREATE EXTERNAL TABLE t STORED AS FASTA|FASTQ|VCF|BAM|INDEXED_VCF|INDEXED_BAM [OPTIONS (compression gzip)] LOCATION '<path>'\n"
             "  functions: fasta_scan fastq_scan vcf_scan bam_scan ('<path>'[, 'gzip']); vcf_indexed_scan bam_indexed_scan ('<path>', '<region>');\n"
             "             fastq_quality_histogram('<path>')\n"
             "  queries:   SELECT COUNT(*) FROM <src> [WHERE chrom = 'c' AND pos >= a AND pos <= b | vcf_region_filter('r', chrom) | bam_region_filter('r', reference, start, end)]\n"
             "             SELECT reference, COUNT(*) FROM <bam> WHERE flag & M = V AND CAST(mapping_quality AS INT) >= q GROUP BY reference\n"
             "             SET exon.vcf_parse_info = true; SELECT filter, AVG(qual), COUNT(*) FROM <vcf> WHERE info.\"AF\" > 0.01 GROUP BY filter\n");
      return 0;
    } else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
  }
  try {
    for (const auto& c : commands) exec_script(se, c);
    for (const auto& f : files) {
      std::ifstream in(f);
      if (!in) throw Err("cannot read " + f);
      std::stringstream ss;
      ss << in.rdbuf();
      exec_script(se, ss.str());
    }
  } catch (const std::exception& e) {
    fprintf(stderr, "Error: %s\n", e.what());
    return 1;
  }
  return 0;
}
