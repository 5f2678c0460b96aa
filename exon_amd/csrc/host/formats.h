// formats.h -- native record decoders + device-layout array builders for VCF / BAM / FASTQ / FASTA.
//
// C++ counterparts of the reference's per-format crates, emitting the columns the GPU kernels consume
// (dictionary ids instead of Utf8 keys, raw u8 mapq, Arrow validity bitmaps) instead of the reference's
// string-heavy Arrow layout:
//   VCFConfig / VCFArrayBuilder / VCFBatchReader
//       exon-vcf/src/config.rs:23-64, array_builder/lazy_array_builder.rs:50-495 (append: :153-448),
//       array_builder/info_builder.rs:152-309 (typed INFO field), async_batch_stream.rs:80-109 (read_batch)
//   BAMConfig / BAMArrayBuilder / BAMBatchReader
//       exon-bam/src/config.rs:21-77 (default batch 8096 there; 8192 here as everywhere else),
//       array_builder.rs:102-218, batch_reader.rs:44-108, indexed_async_batch_stream.rs:35-87 (alignment_end)
//   FASTQConfig / FASTQArrayBuilder / FASTQBatchReader   exon-fastq/src/{config,array_builder,batch_reader}.rs
//   FASTAConfig / FASTAArrayBuilder / FASTABatchReader   exon-fasta/src/{config,array_builder,batch_reader}.rs
// Record syntax itself is noodles' (VCF 4.x text, BAM spec section 4.2, 4-line FASTQ, '>' FASTA).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "arrow_build.h"
#include "bgzf_index.h"
#include "decimal_f32.h"
#include "io.h"
#include "parallel.h"
#include "raw_batch.h"
#include "region.h"
#include "vcf_text.h"

namespace exon {

constexpr int64_t DEFAULT_BATCH_SIZE = 8 * 1024;  // exon-common/src/lib.rs:27

struct RegionFilter {  // pushed-down vcf_region_filter / bam_region_filter (per-record interval hit)
  bool active = false;
  Region region;
  bool use_index = false;  // plan BGZF chunks from <file>.tbi / <file>.bai instead of scanning the whole file
};

// Where records come from: the whole (possibly compressed) stream, or the BGZF chunks an index query returned
// (one PartitionedFile per chunk in the reference: indexed_file/indexed_bgzf_file.rs:129-155).
struct RecordSource {
  virtual ~RecordSource() = default;
  virtual bool next_record() = 0;  // false = no further record may start
  virtual bool read_line(std::string* line) = 0;
  virtual bool read_exact(uint8_t* dst, size_t n) = 0;
  virtual size_t chunk_index() const { return 0; }  // which index chunk the next record comes from (ChunkSource)
};
struct StreamSource : RecordSource {
  BufReader r;
  StreamSource(const std::string& path, Compression c, int threads = 0) : r(open_source(path, c, threads)) {}
  bool next_record() override { return true; }
  bool read_line(std::string* line) override { return r.read_line(line); }
  bool read_exact(uint8_t* dst, size_t n) override { return r.read_exact(dst, n); }
};
struct ChunkSource : RecordSource {
  BgzfReader r;
  std::vector<Chunk> chunks;
  size_t ci = 0;
  bool in_chunk = false;
  ChunkSource(const std::string& path, std::vector<Chunk> c) : r(path), chunks(std::move(c)) {}
  bool next_record() override {
    for (;;) {
      if (ci >= chunks.size()) return false;
      if (!in_chunk) {
        r.seek(chunks[ci].start);
        in_chunk = true;
      }
      if (r.tell() < chunks[ci].end) return true;  // a record may start anywhere before the chunk's end
      ++ci;
      in_chunk = false;
    }
  }
  bool read_line(std::string* line) override { return r.read_line(line); }
  bool read_exact(uint8_t* dst, size_t n) override { return r.read_exact(dst, n); }
  size_t chunk_index() const override { return ci; }
};

// ======================================================================================================
// VCF
// ======================================================================================================
struct VCFHeader {
  std::vector<std::string> contigs;                          // ##contig order = chrom dictionary order
  std::vector<std::string> filters;                          // ##FILTER ids
  std::vector<std::pair<std::string, std::string>> infos;    // (ID, "Number|Type")
  std::vector<std::pair<std::string, std::string>> formats;  // ##FORMAT lines, same form
  std::vector<std::string> samples;
};

struct VCFConfig {
  int64_t batch_size = DEFAULT_BATCH_SIZE;
  int threads = 0;  // decode threads: 0 = all host cores (EXON_HIP_DECODE_THREADS), 1 = sequential reader
  bool defer_decode = false;  // the caller will take the byte stream (GPU-side parsing): start no parse pipeline
  std::string info_field;  // exon.vcf_parse_info=true + SELECT info."<F>": Number=1 Float/Integer field -> f32 column
  RegionFilter filter;
  // EXON_HIP_REFERENCE_QUIRKS=1: reproduce IndexedAsyncBatchStream::read_batch as written
  // (exon-vcf/src/indexed_async_batch_stream.rs:118-166) on indexed scans: one stream per index chunk; a batch takes records
  // until `batch_size` of them hit the region, then appends up to `batch_size` FURTHER records of the chunk UNFILTERED
  // (:143-154).  Off (default): every record is tested -- the documented semantics of vcf_region_filter.
  bool reference_tail_quirk = false;
  uint64_t projection = 0;  // EXON_HIP_PROJECT_VCF_*: id / ref / alt behind the default columns (decoded by ONE thread)
};

inline std::string header_attr(const std::string& line, const char* key) {
  const size_t lt = line.find('<');
  if (lt == std::string::npos) return "";
  std::string k = std::string(key) + "=";
  size_t p = lt + 1;
  bool inq = false;
  size_t field_start = p;
  for (size_t i = p; i <= line.size(); ++i) {
    const char ch = i < line.size() ? line[i] : '>';
    if (ch == '"') inq = !inq;
    if ((ch == ',' || ch == '>') && !inq) {
      if (line.compare(field_start, k.size(), k) == 0) {
        std::string v = line.substr(field_start + k.size(), i - field_start - k.size());
        if (v.size() >= 2 && v.front() == '"' && v.back() == '"') v = v.substr(1, v.size() - 2);
        return v;
      }
      field_start = i + 1;
      if (ch == '>') break;
    }
  }
  return "";
}

// Schema of the device-layout VCF batch: chrom (dict), pos, qual, filter (dict of ';'-joined lists,
// "" = empty list), [info.<F>].  Reference schema: exon-core/src/datasources/vcf/schema_builder.rs:85-129.
// One typed INFO field of the scan (exon.vcf_parse_info = true; `info."<F>"`): InfosBuilder builds a child per header INFO
// (exon-vcf/src/array_builder/info_builder.rs:152-309, typing in exon-core/src/datasources/vcf/schema_builder.rs:197-249); a
// scan here builds the ones the query names (up to MAX_INFO_FIELDS, comma-separated in the options), in the device layout:
//   'f'  Number=1 Float -> f32 + validity                'i'  Number=1 Integer -> i32 + validity (exact: never through f32)
//   'b'  Number=0 Flag -> Boolean, true when present, NULL when absent
//   's'  Number=1 String / Character -> dictionary<int32, utf8> + validity
//   'F' / 'I' / 'S'  any other Number (A, R, G, '.', 2, ...) of Float / Integer / String|Character -> List<f32 / i32 /
//        dictionary<int32, utf8>>: items split on ',', an item '.' is a NULL item (info_builder.rs:258-305)
// Key absent, value '.', or INFO itself '.' (the whole struct is NULL then): NULL in every kind.
struct InfoSpec {
  std::string name;
  char kind = 'f';
};
// Fields per scan: the reference builds a child for EVERY header INFO line; a scan here builds the ones the query names
// (projection pushdown).  16 is what the device parsers' by-value key table holds (exon_hip_vcf_columns.infos[16]).
constexpr int MAX_INFO_FIELDS = 16;
inline bool info_kind_is_list(char k) { return k == 'F' || k == 'I' || k == 'S'; }
inline bool info_kind_on_device(char k) { return k == 'f' || k == 'i' || k == 'b'; }  // the others keep the scan on the host

inline std::vector<std::string> split_list(const std::string& s, char sep = ',') {
  std::vector<std::string> out;
  size_t i = 0;
  while (i <= s.size() && !s.empty()) {
    size_t j = s.find(sep, i);
    if (j == std::string::npos) j = s.size();
    if (j > i) out.push_back(s.substr(i, j - i));
    i = j + 1;
  }
  return out;
}

// "AF,DP,DB" + the header's (ID, "Number|Type") list -> typed specs; throws on unknown / list-valued fields
inline std::vector<InfoSpec> resolve_info_specs(const std::string& fields, const std::vector<std::pair<std::string, std::string>>& header_infos) {
  std::vector<InfoSpec> specs;
  for (const std::string& name : split_list(fields)) {
    const std::string* ty = nullptr;
    for (const auto& kv : header_infos)
      if (kv.first == name) ty = &kv.second;
    if (!ty) throw std::runtime_error("INFO field " + name + " is not declared in the header");
    InfoSpec sp;
    sp.name = name;
    if (*ty == "1|Float") sp.kind = 'f';
    else if (*ty == "1|Integer") sp.kind = 'i';  // Int32 in the reference's schema (schema_builder.rs:197-205)
    else if (*ty == "0|Flag") sp.kind = 'b';
    else if (*ty == "1|String" || *ty == "1|Character") sp.kind = 's';
    else {
      // Number other than 0 / 1 -> List<item> (schema_builder.rs:235-249)
      const size_t bar = ty->find('|');
      const std::string number = ty->substr(0, bar), type = bar == std::string::npos ? "" : ty->substr(bar + 1);
      if (number.empty() || number == "0" || number == "1") throw std::runtime_error("INFO field " + name + " has an unsupported Number / Type (" + *ty + ")");
      if (type == "Float") sp.kind = 'F';
      else if (type == "Integer") sp.kind = 'I';
      else if (type == "String" || type == "Character") sp.kind = 'S';
      else throw std::runtime_error("INFO field " + name + " has an unsupported Number / Type (" + *ty + ")");
    }
    specs.push_back(sp);
  }
  if ((int)specs.size() > MAX_INFO_FIELDS) throw std::runtime_error("at most " + std::to_string(MAX_INFO_FIELDS) + " INFO fields per scan");
  return specs;
}

class VCFArrayBuilder : public ExonArrayBuilder {
 public:
  // info_dicts: one dictionary per spec (used by the 's' kind only), owned by the caller like the other dictionaries
  // key_types: the header's INFO / FORMAT value types, needed by the `info` / `formats` text columns (projection bits 8 / 16)
  VCFArrayBuilder(Dictionary* chrom_dict, Dictionary* filter_dict, const std::vector<InfoSpec>& specs, std::vector<Dictionary>* info_dicts, uint64_t projection = 0,
                  const VcfKeyTypes* key_types = nullptr)
      : chrom_dict_(chrom_dict), filter_dict_(filter_dict), specs_(specs), info_dicts_(info_dicts), info_f_(specs.size()),
        info_i_(specs.size()), info_lf_(specs.size()), info_li_(specs.size()), projection_(projection), key_types_(key_types) {
    if ((projection_ & 24) && !key_types_) throw std::runtime_error("the info / formats text columns need the header's key types");
  }

  // one data line (no terminator).  Field rules: lazy_array_builder.rs:159-216.
  void append(const std::string& line) { append(line.data(), line.size()); }
  void append(const char* line, size_t len) {
    const char* f[9];
    size_t fl[9];
    int nf = 0;
    size_t start = 0;
    for (size_t i = 0; i <= len && nf < 9; ++i)
      if (i == len || line[i] == '\t') {
        f[nf] = line + start;
        fl[nf] = i - start;
        ++nf;
        start = i + 1;
      }
    const size_t samples_at = start;  // behind FORMAT's TAB (> len: the record has no samples)
    if (nf < 8) throw std::runtime_error("VCF record has fewer than 8 fields");
    chrom_.append_value(chrom_dict_->lookup_or_insert(f[0], fl[0]));
    // POS: "0" (telomere) has no variant_start -> NULL; anything that is not a number is the reference's parse error
    // (`record.variant_start().transpose()?`, lazy_array_builder.rs:163-168)
    int64_t pos = 0;
    if (!parse_pos(f[1], fl[1], &pos)) throw std::runtime_error("invalid POS '" + std::string(f[1], fl[1]) + "'");
    if (pos > 0) pos_.append_value(pos);
    else pos_.append_null(0);
    // QUAL: '.' -> NULL, else correctly rounded f32 (Rust str::parse::<f32>)
    if (fl[5] == 1 && f[5][0] == '.') qual_.append_null(0.f);
    else qual_.append_value(parse_f32(f[5], fl[5]));
    // FILTER: '.' -> empty list (never NULL); the list is kept as its ';'-joined text, order preserved
    if (fl[6] == 1 && f[6][0] == '.') filter_.append_value(filter_dict_->lookup_or_insert("", 0));
    else filter_.append_value(filter_dict_->lookup_or_insert(f[6], fl[6]));
    if (!specs_.empty()) append_info(f[7], fl[7]);
    if (projection_ & 1) {  // id: List<Utf8>, NULL when there is none (lazy_array_builder.rs:169-180)
      if (fl[2] == 0 || (fl[2] == 1 && f[2][0] == '.')) {
        id_.append_null();
      } else {
        size_t a = 0;
        for (size_t i = 0; i <= fl[2]; ++i)
          if (i == fl[2] || f[2][i] == ';') {
            id_.items.append_value(f[2] + a, i - a);
            a = i + 1;
          }
        id_.close_row();
      }
    }
    if (projection_ & 2) ref_.append_value(f[3], fl[3]);  // :181-190
    if (projection_ & 4) {
      // alt: the reference builds a string of the alternate bases and then appends the LIST without a value (:191-205): every
      // record with alternate bases gets an empty list, one without gets NULL
      if (fl[4] == 0 || (fl[4] == 1 && f[4][0] == '.')) alt_.append_null();
      else alt_.close_row();
    }
    // info / formats as text: the parsed entries printed again, not the fields' bytes (host/vcf_text.h; lazy_array_builder.rs:216-297, :310-423)
    auto pf = [](const char* p, size_t n) { return parse_f32(p, n); };
    if (projection_ & 8) {
      vcf_info_string(f[7], fl[7], *key_types_, pf, &text_);
      info_text_.append_value(text_.data(), text_.size());
    }
    if (projection_ & 16) {
      const bool has = nf >= 9;
      vcf_formats_string(has ? f[8] : nullptr, has ? fl[8] : 0, has && samples_at <= len ? line + samples_at : nullptr,
                         has && samples_at <= len ? len - samples_at : 0, *key_types_, pf, &text_);
      formats_text_.append_value(text_.data(), text_.size());
    }
    ++rows_;
  }

  // last appended record's (chrom id, pos, has_pos): used by the pushed-down region filter
  size_t len() const override { return rows_; }

  std::vector<struct ArrowArray*> finish() override {
    std::vector<struct ArrowArray*> out;
    out.push_back(chrom_.finish(utf8_array(chrom_dict_->names)));
    out.push_back(pos_.finish());
    out.push_back(qual_.finish());
    out.push_back(filter_.finish(utf8_array(filter_dict_->names)));
    for (size_t k = 0; k < specs_.size(); ++k) {
      if (specs_[k].kind == 'f') out.push_back(info_f_[k].finish());
      else if (specs_[k].kind == 'i') out.push_back(info_i_[k].finish());
      else if (specs_[k].kind == 's') out.push_back(info_i_[k].finish(utf8_array((*info_dicts_)[k].names)));
      else if (specs_[k].kind == 'F') out.push_back(info_lf_[k].finish());
      else if (specs_[k].kind == 'I') out.push_back(info_li_[k].finish());
      else if (specs_[k].kind == 'S') out.push_back(info_li_[k].finish(utf8_array((*info_dicts_)[k].names)));
      else {  // Flag -> Boolean: value true where present
        struct ArrowArray* a = static_cast<struct ArrowArray*>(malloc(sizeof *a));
        make_boolean(a, info_i_[k].valid, info_i_[k].valid);
        info_i_[k].values.clear();
        info_i_[k].valid.clear();
        out.push_back(a);
      }
    }
    if (projection_ & 1) out.push_back(id_.finish());
    if (projection_ & 2) out.push_back(ref_.finish());
    if (projection_ & 4) out.push_back(alt_.finish());
    if (projection_ & 8) out.push_back(info_text_.finish());
    if (projection_ & 16) out.push_back(formats_text_.finish());
    rows_ = 0;
    return out;
  }

  void reserve(size_t rows) {
    for (auto* v : {&chrom_, &filter_}) { v->values.reserve(rows); v->valid.reserve(rows); }
    pos_.values.reserve(rows); pos_.valid.reserve(rows);
    qual_.values.reserve(rows); qual_.valid.reserve(rows);
    for (size_t k = 0; k < specs_.size(); ++k) {
      if (specs_[k].kind == 'f') { info_f_[k].values.reserve(rows); info_f_[k].valid.reserve(rows); }
      else if (!info_kind_is_list(specs_[k].kind)) { info_i_[k].values.reserve(rows); info_i_[k].valid.reserve(rows); }
    }
  }
  // raw column vectors (parallel decoder: slabs are parsed with slab-local dictionaries, then re-keyed)
  PrimitiveBuilder<int32_t>& chrom_ids() { return chrom_; }
  PrimitiveBuilder<int32_t>& filter_ids() { return filter_; }
  PrimitiveBuilder<int64_t>& positions() { return pos_; }
  PrimitiveBuilder<float>& quals() { return qual_; }
  const std::vector<InfoSpec>& info_specs() const { return specs_; }
  PrimitiveBuilder<float>& info_f32(size_t k) { return info_f_[k]; }      // 'f'
  PrimitiveBuilder<int32_t>& info_i32(size_t k) { return info_i_[k]; }    // 'i': values; 's': dictionary ids; 'b': 1 where present

  // INFO Type=Integer value: Rust's `str::parse::<i32>` ([+-] digits, the whole field, no blanks); anything else -- and a
  // value outside int32 -- is the reference's parse error for the record (noodles `Info::get(...).transpose()?`)
  static int32_t parse_i32(const char* p, size_t n) {
    size_t i = 0;
    bool neg = false;
    if (i < n && (p[i] == '-' || p[i] == '+')) neg = p[i++] == '-';
    if (i == n) throw std::runtime_error("invalid INFO integer '" + std::string(p, n) + "'");
    int64_t v = 0;
    for (; i < n; ++i) {
      if (p[i] < '0' || p[i] > '9') throw std::runtime_error("invalid INFO integer '" + std::string(p, n) + "'");
      v = v * 10 + (p[i] - '0');
      if (v > (int64_t)INT32_MAX + 1) throw std::runtime_error("INFO integer out of the int32 range '" + std::string(p, n) + "'");
    }
    if (neg) v = -v;
    if (v > INT32_MAX || v < INT32_MIN) throw std::runtime_error("INFO integer out of the int32 range '" + std::string(p, n) + "'");
    return (int32_t)v;
  }

  // Rust's usize::from_str -- what noodles-vcf 0.70 parses POS with: the crate has no number parser of its own among its
  // dependencies (the reference's Cargo.lock:3915-3930), and core's FromStr for unsigned integers takes an optional '+' and then
  // digits only -- so "+5" is 5, while "-5", "", "+", " 7", "1e3" are errors
  static bool parse_pos(const char* p, size_t n, int64_t* out) {
    if (n && p[0] == '+') ++p, --n;
    if (n == 0 || n > 18) return false;
    int64_t v = 0;
    for (size_t i = 0; i < n; ++i) {
      if (p[i] < '0' || p[i] > '9') return false;
      v = v * 10 + (p[i] - '0');
    }
    *out = v;
    return true;
  }

  // Correctly rounded decimal -> f32 (what Rust's str::parse::<f32> guarantees).  Fast path (Clinger): a
  // mantissa below 2^24 and a power of ten up to 10^10 are both exact in f32, so ONE IEEE multiply/divide is
  // correctly rounded; anything else (long mantissas, big exponents, inf/nan spellings) goes to strtof.
  static float parse_f32(const char* p, size_t n) {
    {
      static const float P10[11] = {1e0f, 1e1f, 1e2f, 1e3f, 1e4f, 1e5f, 1e6f, 1e7f, 1e8f, 1e9f, 1e10f};
      size_t i = 0;
      bool neg = false;
      if (i < n && (p[i] == '-' || p[i] == '+')) neg = p[i++] == '-';
      uint32_t mant = 0;
      int digits = 0, frac = 0;
      bool ok = true, seen = false;
      for (; i < n && p[i] >= '0' && p[i] <= '9'; ++i) {
        seen = true;
        if (mant || p[i] != '0') { mant = mant * 10 + (uint32_t)(p[i] - '0'); if (++digits > 7) { ok = false; break; } }
      }
      if (ok && i < n && p[i] == '.') {
        for (++i; i < n && p[i] >= '0' && p[i] <= '9'; ++i) {
          seen = true;
          ++frac;
          if (mant || p[i] != '0') { mant = mant * 10 + (uint32_t)(p[i] - '0'); if (++digits > 7) { ok = false; break; } }
        }
      }
      int e10 = 0;
      if (ok && seen && i < n && (p[i] == 'e' || p[i] == 'E')) {
        ++i;
        bool eneg = false;
        if (i < n && (p[i] == '-' || p[i] == '+')) eneg = p[i++] == '-';
        int ed = 0;
        for (; i < n && p[i] >= '0' && p[i] <= '9' && ed < 4; ++i, ++ed) e10 = e10 * 10 + (p[i] - '0');
        if (ed == 0) ok = false;
        if (eneg) e10 = -e10;
      }
      if (ok && seen && i == n && mant < (1u << 24)) {
        const int e = e10 - frac;
        if (mant == 0) return neg ? -0.0f : 0.0f;
        if (e == 0) return neg ? -(float)mant : (float)mant;
        if (e < 0 && e >= -10) { const float v = (float)mant / P10[-e]; return neg ? -v : v; }
        if (e > 0 && e <= 10) { const float v = (float)mant * P10[e]; return neg ? -v : v; }
      }
    }
    // Everything else: the Eisel-Lemire path the GPU parser uses (<= 19 significant digits), then strtof for what is left
    // -- but only for text Rust's grammar accepts: [sign] (inf | infinity | nan | digits [. digits] [e [sign] digits]), the
    // whole field, no hex floats, no leading blanks, any length, independent of the process locale's decimal point.
    uint32_t bits;
    if (n <= (size_t)INT32_MAX && dec::parse_f32(p, (int)n, &bits)) {
      float v;
      memcpy(&v, &bits, 4);
      return v;
    }
    const std::string tmp(p, n);
    size_t i = 0;
    if (i < n && (p[i] == '+' || p[i] == '-')) ++i;
    auto ieq = [&](const char* w) {
      const size_t wl = strlen(w);
      if (n - i != wl) return false;
      for (size_t k = 0; k < wl; ++k)
        if ((p[i + k] | 0x20) != w[k]) return false;
      return true;
    };
    bool ok = ieq("inf") || ieq("infinity") || ieq("nan");
    if (!ok) {
      ok = i < n;
      for (size_t k = i; k < n && ok; ++k) ok = (p[k] >= '0' && p[k] <= '9') || p[k] == '.' || p[k] == 'e' || p[k] == 'E' || p[k] == '+' || p[k] == '-';
    }
    char* end = nullptr;
    const float v = ok ? strtof(tmp.c_str(), &end) : 0.f;
    if (!ok || end != tmp.c_str() + n) throw std::runtime_error("invalid float '" + tmp + "'");
    return v;
  }

 private:
  // INFO '.' -> NULL struct (every field NULL); key absent or value '.' -> NULL field (info_builder.rs:152-309).  One pass over
  // the ';'-separated entries; the FIRST occurrence of a key wins.
  void append_info(const char* p, size_t n) {
    const size_t K = specs_.size();
    bool seen[MAX_INFO_FIELDS] = {};
    auto null_of = [&](size_t k) {
      const char kind = specs_[k].kind;
      if (kind == 'f') info_f_[k].append_null(0.f);
      else if (kind == 'F') info_lf_[k].append_null();
      else if (kind == 'I' || kind == 'S') info_li_[k].append_null();
      else info_i_[k].append_null(0);
    };
    if (!(n == 1 && p[0] == '.')) {
      size_t i = 0;
      while (i < n) {
        size_t j = i;
        while (j < n && p[j] != ';') ++j;
        size_t eq = i;
        while (eq < j && p[eq] != '=') ++eq;
        for (size_t k = 0; k < K; ++k) {
          const std::string& name = specs_[k].name;
          if (seen[k] || eq - i != name.size() || memcmp(p + i, name.data(), name.size()) != 0) continue;
          const char* v = eq < j ? p + eq + 1 : p + j;
          const size_t vl = eq < j ? j - eq - 1 : 0;
          const bool missing = eq < j && (vl == 0 || (vl == 1 && v[0] == '.'));
          const char kind = specs_[k].kind;
          seen[k] = true;
          if (kind == 'b') {
            info_i_[k].append_value(1);  // a Flag is true by being there
          } else if (eq < j && !missing) {
            if (kind == 'f') info_f_[k].append_value(parse_f32(v, vl));
            else if (kind == 'i') info_i_[k].append_value(parse_i32(v, vl));
            else if (kind == 's') info_i_[k].append_value((*info_dicts_)[k].lookup_or_insert(v, vl));
            else {  // list: items split on ','; '.' (or an empty item) is a NULL item
              size_t a = 0;
              while (a <= vl) {
                size_t e = a;
                while (e < vl && v[e] != ',') ++e;
                const bool dot = e == a || (e - a == 1 && v[a] == '.');
                if (kind == 'F') {
                  if (dot) info_lf_[k].items.append_null(0.f);
                  else info_lf_[k].items.append_value(parse_f32(v + a, e - a));
                } else if (kind == 'I') {
                  if (dot) info_li_[k].items.append_null(0);
                  else info_li_[k].items.append_value(parse_i32(v + a, e - a));
                } else {
                  if (dot) info_li_[k].items.append_null(0);
                  else info_li_[k].items.append_value((*info_dicts_)[k].lookup_or_insert(v + a, e - a));
                }
                a = e + 1;
              }
              if (kind == 'F') info_lf_[k].close_row();
              else info_li_[k].close_row();
            }
          } else {
            null_of(k);  // `key=.` / bare key of a valued field: present but missing -> NULL
          }
        }
        i = j + 1;
      }
    }
    for (size_t k = 0; k < K; ++k)
      if (!seen[k]) null_of(k);
  }

  Dictionary *chrom_dict_, *filter_dict_;
  std::vector<InfoSpec> specs_;
  std::vector<Dictionary>* info_dicts_;
  PrimitiveBuilder<int32_t> chrom_, filter_;
  PrimitiveBuilder<int64_t> pos_;
  PrimitiveBuilder<float> qual_;
  std::vector<PrimitiveBuilder<float>> info_f_;
  std::vector<PrimitiveBuilder<int32_t>> info_i_;
  std::vector<ListBuilder<float>> info_lf_;    // 'F'
  std::vector<ListBuilder<int32_t>> info_li_;  // 'I' values, 'S' dictionary ids
  uint64_t projection_ = 0;
  const VcfKeyTypes* key_types_ = nullptr;
  ListUtf8Builder id_, alt_;
  Utf8Builder ref_, info_text_, formats_text_;
  std::string text_;
  size_t rows_ = 0;
};

// IndexedAsyncBatchStream::filter (exon-vcf/src/indexed_async_batch_stream.rs:99-116) on a raw data line
inline bool vcf_region_hit(const char* line, size_t len, const Region& rg) {
  const char* t1 = static_cast<const char*>(memchr(line, '\t', len));
  if (!t1) return false;
  const size_t nl = (size_t)(t1 - line);
  if (nl != rg.name.size() || memcmp(line, rg.name.data(), nl) != 0) return false;
  // the contig matches: now the position is parsed, and a malformed one is an error (`position?`), "0" is no position
  const char* p = t1 + 1;
  const char* end = static_cast<const char*>(memchr(p, '\t', (size_t)(line + len - p)));
  if (!end) end = line + len;
  int64_t pos = 0;
  if (!VCFArrayBuilder::parse_pos(p, (size_t)(end - p), &pos)) throw std::runtime_error("invalid POS '" + std::string(p, (size_t)(end - p)) + "'");
  return pos >= 1 && pos >= rg.start && pos <= rg.end;
}

// one slab of VCF text parsed with slab-local dictionaries (header contigs pre-seeded, so their ids are global)
struct VCFParseCtx {
  std::vector<std::string> contigs;
  std::vector<InfoSpec> info_specs;  // numeric / Flag kinds only: string INFO fields keep the reader sequential
  RegionFilter filter;
};
struct VCFSlab : TextSlab {
  Dictionary chrom_dict, filter_dict;
  std::vector<Dictionary> info_dicts;
  std::unique_ptr<VCFArrayBuilder> b;
  size_t rows = 0;
};
inline void parse_vcf_slab(VCFSlab& s, const void* vctx) {
  const VCFParseCtx& ctx = *static_cast<const VCFParseCtx*>(vctx);
  s.chrom_dict.names = ctx.contigs;
  s.info_dicts.assign(ctx.info_specs.size(), Dictionary());
  s.b.reset(new VCFArrayBuilder(&s.chrom_dict, &s.filter_dict, ctx.info_specs, &s.info_dicts));
  const char* p = s.data();
  const char* end = p + s.len;
  s.b->reserve(s.len / 48 + 16);
  while (p < end) {
    const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
    size_t len = nl ? (size_t)(nl - p) : (size_t)(end - p);
    const char* next = nl ? nl + 1 : end;
    if (len && p[len - 1] == '\r') --len;
    if (len && p[0] != '#' && (!ctx.filter.active || vcf_region_hit(p, len, ctx.filter.region))) s.b->append(p, len);
    p = next;
  }
  s.rows = s.b->len();
}

class VCFBatchReader {
 public:
  VCFBatchReader(const std::string& path, Compression c, VCFConfig cfg) : cfg_(std::move(cfg)) {
    r_.reset(new StreamSource(path, c, cfg_.threads));
    // header (noodles `read_header`): meta lines '##', then '#CHROM ...'
    std::string line;
    while (r_->read_line(&line)) {
      if (line.rfind("##", 0) == 0) {
        if (line.rfind("##contig=", 0) == 0) header.contigs.push_back(header_attr(line, "ID"));
        else if (line.rfind("##FILTER=", 0) == 0) header.filters.push_back(header_attr(line, "ID"));
        else if (line.rfind("##INFO=", 0) == 0)
          header.infos.emplace_back(header_attr(line, "ID"), header_attr(line, "Number") + "|" + header_attr(line, "Type"));
        else if (line.rfind("##FORMAT=", 0) == 0)
          header.formats.emplace_back(header_attr(line, "ID"), header_attr(line, "Number") + "|" + header_attr(line, "Type"));
        continue;
      }
      if (!line.empty() && line[0] == '#') {
        size_t col = 0, start = 0;
        for (size_t i = 0; i <= line.size(); ++i)
          if (i == line.size() || line[i] == '\t') {
            if (col >= 9) header.samples.push_back(line.substr(start, i - start));
            ++col;
            start = i + 1;
          }
        break;
      }
      pending_ = line;  // headerless input: first data line
      has_pending_ = true;
      break;
    }
    for (const auto& c2 : header.contigs) chrom_dict.names.push_back(c2);
    for (const auto& kv : header.infos) key_types.info.emplace(kv.first, VcfKeyTypes::type_char(kv.second.substr(kv.second.find('|') + 1)));  // (the first line of an ID wins)
    for (const auto& kv : header.formats) key_types.format.emplace(kv.first, VcfKeyTypes::type_char(kv.second.substr(kv.second.find('|') + 1)));
    info_specs = resolve_info_specs(cfg_.info_field, header.infos);  // INFO typing: schema_builder.rs:197-249
    info_dicts.assign(info_specs.size(), Dictionary());
    bool string_info = false;
    for (const auto& sp : info_specs) string_info |= !info_kind_on_device(sp.kind);  // dictionary / list kinds: sequential reader
    if (cfg_.filter.active && cfg_.filter.use_index) {
      // get_byte_range_for_file (indexed_bgzf_file.rs:52-112): tabix names -> id -> index.query -> chunks
      const BinningIndex idx = read_tabix(path + ".tbi");
      int id = -1;
      for (size_t i = 0; i < idx.names.size(); ++i)
        if (idx.names[i] == cfg_.filter.region.name) id = (int)i;
      std::vector<Chunk> chunks;
      if (id >= 0) chunks = query_index(idx, id, cfg_.filter.region.start, cfg_.filter.region.end);
      n_chunks = (int)chunks.size();
      planned_chunks = chunks;
      has_pending_ = false;
      r_.reset(new ChunkSource(path, std::move(chunks)));
    } else {
      // multi-threaded decode of the rest of the stream (files of at least a couple of slabs)
      const int threads = cfg_.threads > 0 ? cfg_.threads : decode_threads();
      const long fsize = file_size(path);
      if (threads > 1 && fsize >= (8 << 20) && !cfg_.defer_decode && !string_info && !cfg_.projection) {  // (id / ref / alt: the sequential builder)
        StreamSource* ss = static_cast<StreamSource*>(r_.get());
        std::string carry = has_pending_ ? pending_ + "\n" : std::string();
        has_pending_ = false;
        carry += ss->r.take_buffered();
        pctx_.contigs = header.contigs;
        pctx_.info_specs = info_specs;
        pctx_.filter = cfg_.filter;
        pipe_.reset(new SlabPipeline<VCFSlab>(ss->r.release_source(), std::move(carry), 1, threads,
                                              [](VCFSlab& s, const void* c) { parse_vcf_slab(s, c); }, &pctx_));
      }
    }
  }

  // AsyncBatchStream::read_batch (exon-vcf/src/async_batch_stream.rs:80-109); with a region filter the
  // per-record test of IndexedAsyncBatchStream::filter applies to EVERY record
  // (exon-vcf/src/indexed_async_batch_stream.rs:99-116; see DESIGN.md on the reference's unfiltered tail).
  bool read_batch(struct ArrowArray* out) {
    if (pipe_) return read_batch_parallel(out);
    if (cfg_.reference_tail_quirk && n_chunks >= 0 && cfg_.filter.active) return read_batch_reference_quirk(out);
    VCFArrayBuilder b(&chrom_dict, &filter_dict, info_specs, &info_dicts, cfg_.projection, &key_types);
    std::string line;
    while ((int64_t)b.len() < cfg_.batch_size) {
      if (has_pending_) {
        line.swap(pending_);
        has_pending_ = false;
      } else if (!r_->next_record() || !r_->read_line(&line)) {
        break;
      }
      if (line.empty() || line[0] == '#') continue;
      if (cfg_.filter.active && !vcf_region_hit(line.data(), line.size(), cfg_.filter.region)) continue;
      b.append(line);
    }
    if (b.is_empty()) return false;
    b.try_into_record_batch(out);
    return true;
  }

  // IndexedAsyncBatchStream::read_batch exactly as the reference has it (exon-vcf/src/indexed_async_batch_stream.rs:118-166),
  // one stream per index chunk (indexed_bgzf_file.rs:145-152: one PartitionedFile per chunk): the first loop takes records
  // until batch_size of them passed `filter` or the chunk ends; the second loop then reads up to batch_size MORE records and
  // appends them WITHOUT the test (it reads nothing when the first loop stopped at the chunk's end).  Opt-in only.
  bool read_batch_reference_quirk(struct ArrowArray* out) {
    VCFArrayBuilder b(&chrom_dict, &filter_dict, info_specs, &info_dicts, cfg_.projection, &key_types);
    std::string line;
    for (;;) {  // skip chunks that yield nothing (the reference's stream of such a chunk ends without a batch)
      if (!r_->next_record()) return false;
      const size_t chunk = r_->chunk_index();
      auto next_of_chunk = [&](std::string* l) {  // read_record(): None at the end of THIS chunk
        for (;;) {
          if (!r_->next_record() || r_->chunk_index() != chunk || !r_->read_line(l)) return false;
          if (!l->empty() && (*l)[0] != '#') return true;
        }
      };
      int64_t hits = 0;
      bool ended = false;
      while (hits < cfg_.batch_size) {
        if (!next_of_chunk(&line)) {
          ended = true;
          break;
        }
        if (vcf_region_hit(line.data(), line.size(), cfg_.filter.region)) {
          b.append(line);
          ++hits;
        }
      }
      for (int64_t i = 0; i < cfg_.batch_size && !ended; ++i) {
        if (!next_of_chunk(&line)) break;
        b.append(line);  // unfiltered: indexed_async_batch_stream.rs:143-154
      }
      if (!b.is_empty()) break;
    }
    b.try_into_record_batch(out);
    return true;
  }

  // Everything after the header as a raw byte stream (for the GPU-side parser); `carry` receives the bytes already
  // buffered.  Only valid before the first read_batch / read_raw and when no parse pipeline / index source is active.
  std::unique_ptr<ByteSource> take_stream(std::string* carry) {
    if (pipe_ || n_chunks >= 0) return nullptr;
    StreamSource* ss = static_cast<StreamSource*>(r_.get());
    *carry = has_pending_ ? pending_ + "\n" : std::string();
    has_pending_ = false;
    *carry += ss->r.take_buffered();
    return ss->r.release_source();
  }
  const VCFConfig& config() const { return cfg_; }
  // Uncompressed offset of the first data line (header length), for callers that re-open the file themselves
  // (GPU-side BGZF inflate).  -1 when it is not known (headerless input, parse pipeline or index source active).
  int64_t data_offset() const {
    if (pipe_ || n_chunks >= 0 || has_pending_) return -1;
    return (int64_t)static_cast<StreamSource*>(r_.get())->r.consumed();
  }

  // Parallel mode only: the rest of the current slab as raw columns (no Arrow materialisation); false when the
  // reader is sequential (use read_batch) or the input is exhausted (*end = true).
  bool read_raw(RawBatch* out, bool* end) {
    *end = false;
    if (!pipe_) return false;
    if (!next_slab()) {
      *end = true;
      return false;
    }
    const size_t o = cur_pos_, n = cur_->rows - o;
    VCFArrayBuilder& b = *cur_->b;
    out->rows = (int64_t)n;
    out->cols.clear();
    out->cols.push_back({b.chrom_ids().values.data() + o, nullptr, 4});
    out->cols.push_back({b.positions().values.data() + o, b.positions().valid.data() + o, 8});
    out->cols.push_back({b.quals().values.data() + o, b.quals().valid.data() + o, 4});
    out->cols.push_back({b.filter_ids().values.data() + o, nullptr, 4});
    for (size_t k = 0; k < info_specs.size(); ++k) {  // 'f': f32 values; 'b': the 0 / 1 words of the presence column
      if (info_specs[k].kind == 'f') out->cols.push_back({b.info_f32(k).values.data() + o, b.info_f32(k).valid.data() + o, 4});
      else out->cols.push_back({b.info_i32(k).values.data() + o, b.info_i32(k).valid.data() + o, 4});
    }
    cur_pos_ = cur_->rows;
    return true;
  }

  void schema(struct ArrowSchema* out) const {
    std::vector<struct ArrowSchema*> kids = {new_field("i", "chrom", false, new_field("u", "", false)),
                                             new_field("l", "pos", true), new_field("f", "qual", true),
                                             new_field("i", "filter", false, new_field("u", "", false))};
    for (const auto& sp : info_specs) {
      const std::string name = "info." + sp.name;
      if (sp.kind == 'f') kids.push_back(new_field("f", name.c_str(), true));
      else if (sp.kind == 'i') kids.push_back(new_field("i", name.c_str(), true));
      else if (sp.kind == 'b') kids.push_back(new_field("b", name.c_str(), true));
      else if (sp.kind == 'F') kids.push_back(new_list_field("f", name.c_str()));
      else if (sp.kind == 'I') kids.push_back(new_list_field("i", name.c_str()));
      else if (sp.kind == 'S') kids.push_back(new_list_field("i", name.c_str(), new_field("u", "", false)));
      else kids.push_back(new_field("i", name.c_str(), true, new_field("u", "", false)));
    }
    if (cfg_.projection & 1) kids.push_back(new_list_field("u", "id"));
    if (cfg_.projection & 2) kids.push_back(new_field("u", "ref", false));
    if (cfg_.projection & 4) kids.push_back(new_list_field("u", "alt"));
    if (cfg_.projection & 8) kids.push_back(new_field("u", "info", true));      // schema_builder.rs:119-121: both nullable Utf8
    if (cfg_.projection & 16) kids.push_back(new_field("u", "formats", true));  // (the builder never appends a NULL)
    make_schema(out, "+s", "", false, kids);
  }

  VCFHeader header;
  VcfKeyTypes key_types;  // value types of the header's INFO / FORMAT keys
  Dictionary chrom_dict, filter_dict;
  std::vector<InfoSpec> info_specs;    // the typed INFO fields of this scan (scan columns 4 ..)
  std::vector<Dictionary> info_dicts;  // dictionaries of the 's' kind (same index as info_specs)
  int n_chunks = -1;  // index chunks planned (-1: not an indexed scan)
  std::vector<Chunk> planned_chunks;  // ... and the chunks themselves (the GPU decode path ships exactly these blocks)

 private:
  // emit up to batch_size rows of the current slab (re-keyed to the reader's dictionaries)
  // advance to the next non-empty slab and re-key it; false at end of input
  bool next_slab() {
    while (!cur_ || cur_pos_ >= cur_->rows) {
      cur_ = pipe_->next();
      if (!cur_) return false;
      cur_pos_ = 0;
      // slab-local ids -> global ids (header contigs keep their ids; unseen contigs / FILTER lists are interned
      // in file order, exactly as the sequential reader would)
      std::vector<int32_t> cmap(cur_->chrom_dict.names.size()), fmap(cur_->filter_dict.names.size());
      for (size_t i = 0; i < cmap.size(); ++i)
        cmap[i] = i < header.contigs.size() ? (int32_t)i : chrom_dict.lookup_or_insert(cur_->chrom_dict.names[i].data(), cur_->chrom_dict.names[i].size());
      for (int32_t& v : cur_->b->chrom_ids().values) v = cmap[(size_t)v];
      // FILTER lists must be interned in order of first appearance in the file, not of the slab dictionary
      std::vector<int32_t>& fv = cur_->b->filter_ids().values;
      std::fill(fmap.begin(), fmap.end(), -1);
      for (int32_t& v : fv) {
        int32_t& g = fmap[(size_t)v];
        if (g < 0) g = filter_dict.lookup_or_insert(cur_->filter_dict.names[(size_t)v].data(), cur_->filter_dict.names[(size_t)v].size());
        v = g;
      }
    }
    return true;
  }

  bool read_batch_parallel(struct ArrowArray* out) {
    if (!next_slab()) return false;
    const size_t n = std::min<size_t>((size_t)cfg_.batch_size, cur_->rows - cur_pos_), o = cur_pos_;
    auto slice = [&](auto& pb, int elem, struct ArrowArray* dict) {
      struct ArrowArray* a = static_cast<struct ArrowArray*>(malloc(sizeof *a));
      std::vector<uint8_t> valid(pb.valid.begin() + (long)o, pb.valid.begin() + (long)(o + n));
      make_primitive(a, reinterpret_cast<const uint8_t*>(pb.values.data()) + o * (size_t)elem, (int64_t)n, elem, valid, dict);
      return a;
    };
    std::vector<struct ArrowArray*> kids;
    kids.push_back(slice(cur_->b->chrom_ids(), 4, utf8_array(chrom_dict.names)));
    kids.push_back(slice(cur_->b->positions(), 8, nullptr));
    kids.push_back(slice(cur_->b->quals(), 4, nullptr));
    kids.push_back(slice(cur_->b->filter_ids(), 4, utf8_array(filter_dict.names)));
    for (size_t k = 0; k < info_specs.size(); ++k) {
      if (info_specs[k].kind == 'f') {
        kids.push_back(slice(cur_->b->info_f32(k), 4, nullptr));
      } else if (info_specs[k].kind == 'i') {
        kids.push_back(slice(cur_->b->info_i32(k), 4, nullptr));
      } else {  // Flag (string kinds never reach the parallel reader)
        struct ArrowArray* a = static_cast<struct ArrowArray*>(malloc(sizeof *a));
        const auto& pb = cur_->b->info_i32(k);
        std::vector<uint8_t> valid(pb.valid.begin() + (long)o, pb.valid.begin() + (long)(o + n));
        make_boolean(a, valid, valid);
        kids.push_back(a);
      }
    }
    make_struct(out, (int64_t)n, std::move(kids));
    cur_pos_ += n;
    return true;
  }

  std::unique_ptr<RecordSource> r_;
  VCFConfig cfg_;
  std::string pending_;
  bool has_pending_ = false;
  VCFParseCtx pctx_;
  std::unique_ptr<VCFSlab> cur_;
  size_t cur_pos_ = 0;
  std::unique_ptr<SlabPipeline<VCFSlab>> pipe_;  // declared last: destroyed (threads joined) first
};

// ======================================================================================================
// BAM
// ======================================================================================================
struct BAMConfig {
  int64_t batch_size = DEFAULT_BATCH_SIZE;
  int threads = 0;  // BGZF inflate threads (0 = all host cores)
  RegionFilter filter;  // bam_region_filter: SemiLazyRecord::intersects
  uint64_t projection = 0;  // EXON_HIP_PROJECT_BAM_*: name / cigar / sequence / quality_scores behind the default columns
};

// device-layout BAM batch: flag i32, mapq u8? (255 -> NULL), reference dict? (-1 -> NULL), start i64?, end i64?
class BAMArrayBuilder : public ExonArrayBuilder {
 public:
  explicit BAMArrayBuilder(const std::vector<std::string>* ref_names, uint64_t projection = 0) : ref_names_(ref_names), projection_(projection) {}
  // the record's variable part (exon-bam/src/array_builder.rs:105-201); rec = the bytes behind block_size
  void append_text(const uint8_t* rec, size_t n) {
    if (!projection_) return;
    const uint32_t l_name = rec[8];
    uint16_t n_cigar;
    int32_t l_seq;
    memcpy(&n_cigar, rec + 12, 2);
    memcpy(&l_seq, rec + 16, 4);
    const uint8_t* c = rec + 32 + l_name;
    const uint8_t* s = c + 4u * n_cigar;
    const uint8_t* q = s + ((size_t)l_seq + 1) / 2;
    if (l_seq < 0 || (size_t)(q - rec) + (size_t)l_seq > n) throw std::runtime_error("corrupt BAM record");
    if (projection_ & 1) {
      if (l_name == 2 && rec[32] == '*') name_.append_null();  // noodles: "*" is a missing name
      else name_.append_value(reinterpret_cast<const char*>(rec + 32), l_name ? l_name - 1 : 0);
    }
    if (projection_ & 2) {
      std::string t;
      for (uint16_t k = 0; k < n_cigar; ++k) {
        uint32_t op;
        memcpy(&op, c + 4u * k, 4);
        t += std::to_string(op >> 4);
        t += (op & 0xF) < 9 ? "MIDNSHP=X"[op & 0xF] : '?';
      }
      cigar_.append_value(t);
    }
    if (projection_ & 4) {
      std::string t((size_t)l_seq, '=');
      for (int32_t i = 0; i < l_seq; ++i) t[(size_t)i] = "=ACMGRSVTWYHKDBN"[(s[i >> 1] >> ((i & 1) ? 0 : 4)) & 0xF];
      seq_.append_value(t);
    }
    if (projection_ & 8) {
      for (int32_t i = 0; i < l_seq; ++i) qual_.items.append_value((int64_t)(int8_t)q[i]);
      qual_.close_row();
    }
  }
  // the same columns from a SAM line's fields (exon-sam/src/array_builder.rs:101-185 over noodles' RecordBuf): QNAME "*" is no
  // name; the CIGAR is printed again op by op ("*" -> ""); SEQ "*" -> ""; QUAL "*" -> an empty list, else Phred = char - 33
  void append_sam_text(const char* name, size_t nn, const char* cigar, size_t nc, const char* seq, size_t ns, const char* qual, size_t nq) {
    if (projection_ & 1) {
      if (nn == 1 && name[0] == '*') name_.append_null();
      else name_.append_value(name, nn);
    }
    if (projection_ & 2) {
      std::string t;
      if (!(nc == 1 && cigar[0] == '*')) {
        uint64_t num = 0;
        bool digits = false;
        for (size_t i = 0; i < nc; ++i) {
          const char ch = cigar[i];
          if (ch >= '0' && ch <= '9') {
            num = num * 10 + (uint64_t)(ch - '0');
            digits = true;
          } else {
            if (!digits || !strchr("MIDNSHP=X", ch)) throw std::runtime_error("invalid CIGAR '" + std::string(cigar, nc) + "'");
            t += std::to_string(num);
            t += ch;
            num = 0;
            digits = false;
          }
        }
        if (digits) throw std::runtime_error("invalid CIGAR '" + std::string(cigar, nc) + "'");
      }
      cigar_.append_value(t);
    }
    if (projection_ & 4) {
      if (ns == 1 && seq[0] == '*') seq_.append_value("", 0);
      else seq_.append_value(seq, ns);
    }
    if (projection_ & 8) {
      if (!(nq == 1 && qual[0] == '*'))
        for (size_t i = 0; i < nq; ++i) {
          if (qual[i] < 33 || qual[i] > 126) throw std::runtime_error("invalid quality score character");
          qual_.items.append_value((int64_t)(qual[i] - 33));
        }
      qual_.close_row();
    }
  }
  void append(int32_t flag, int32_t ref_id, int64_t pos0, int mapq, int64_t ref_len) {
    flag_.append_value(flag);  // array_builder.rs:114-117: raw u16 bits as Int32
    if (mapq == 255) mapq_.append_null(255);  // :136-143 (reference: decimal string, NULL when missing)
    else mapq_.append_value((uint8_t)mapq);
    if (ref_id < 0) ref_.append_null(-1);  // :118-127
    else ref_.append_value(ref_id);
    if (pos0 < 0) {
      start_.append_null(0);
      end_.append_null(0);
    } else {
      start_.append_value(pos0 + 1);            // 1-based
      end_.append_value(pos0 + 1 + ref_len - 1);  // alignment_end = start + reference length - 1
    }
    ++rows_;
  }
  size_t len() const override { return rows_; }
  std::vector<struct ArrowArray*> finish() override {
    rows_ = 0;
    std::vector<struct ArrowArray*> out = {flag_.finish(), mapq_.finish(), ref_.finish(utf8_array(*ref_names_)), start_.finish(), end_.finish()};
    if (projection_ & 1) out.push_back(name_.finish());
    if (projection_ & 2) out.push_back(cigar_.finish());
    if (projection_ & 4) out.push_back(seq_.finish());
    if (projection_ & 8) {
      // (quality_scores items are never NULL: the child carries no validity bitmap)
      qual_.items.valid.clear();
      out.push_back(qual_.finish());
    }
    return out;
  }

 private:
  const std::vector<std::string>* ref_names_;
  PrimitiveBuilder<int32_t> flag_, ref_;
  PrimitiveBuilder<uint8_t> mapq_;
  PrimitiveBuilder<int64_t> start_, end_;
  uint64_t projection_ = 0;
  Utf8Builder name_, cigar_, seq_;
  ListBuilder<int64_t> qual_;
  size_t rows_ = 0;
};

class BAMBatchReader {
 public:
  BAMBatchReader(const std::string& path, BAMConfig cfg) : cfg_(std::move(cfg)) {
    r_.reset(new StreamSource(path, Compression::Gzip, cfg_.threads));
    uint8_t magic[4];
    if (!r_->read_exact(magic, 4) || memcmp(magic, "BAM\1", 4) != 0) throw std::runtime_error("not a BAM file: " + path);
    const int32_t l_text = read_i32();
    header_text.resize((size_t)l_text);
    if (l_text && !r_->read_exact(reinterpret_cast<uint8_t*>(&header_text[0]), (size_t)l_text)) throw std::runtime_error("truncated BAM header");
    const int32_t n_ref = read_i32();
    for (int32_t i = 0; i < n_ref; ++i) {
      const int32_t l_name = read_i32();
      std::string name((size_t)l_name, '\0');
      if (!r_->read_exact(reinterpret_cast<uint8_t*>(&name[0]), (size_t)l_name)) throw std::runtime_error("truncated BAM reference");
      name.resize(strlen(name.c_str()));
      ref_names.push_back(name);
      ref_lengths.push_back(read_i32());
    }
    if (cfg_.filter.active) {
      region_ref_id_ = -2;
      for (size_t i = 0; i < ref_names.size(); ++i)
        if (ref_names[i] == cfg_.filter.region.name) region_ref_id_ = (int32_t)i;
      if (cfg_.filter.use_index) {  // .bai -> chunks (indexed_bgzf_file.rs:87-107)
        const BinningIndex idx = read_bai(path + ".bai");
        std::vector<Chunk> chunks;
        if (region_ref_id_ >= 0) chunks = query_index(idx, region_ref_id_, cfg_.filter.region.start, cfg_.filter.region.end);
        n_chunks = (int)chunks.size();
        planned_chunks = chunks;
        r_.reset(new ChunkSource(path, std::move(chunks)));
      }
    }
  }

  const BAMConfig& config() const { return cfg_; }
  // uncompressed offset of the first record (header length), for the GPU-side inflate + record splitting
  int64_t data_offset() const { return n_chunks >= 0 ? -1 : (int64_t)static_cast<StreamSource*>(r_.get())->r.consumed(); }

  bool read_batch(struct ArrowArray* out) {
    BAMArrayBuilder b(&ref_names, cfg_.projection);
    std::vector<uint8_t> rec;
    while ((int64_t)b.len() < cfg_.batch_size) {
      uint8_t szb[4];
      if (!r_->next_record() || !r_->read_exact(szb, 4)) break;
      int32_t block;
      memcpy(&block, szb, 4);
      if (block < 32) throw std::runtime_error("corrupt BAM record");
      rec.resize((size_t)block);
      if (!r_->read_exact(rec.data(), rec.size())) throw std::runtime_error("truncated BAM record");
      int32_t ref_id, pos;
      uint16_t n_cigar, flag;
      memcpy(&ref_id, &rec[0], 4);
      memcpy(&pos, &rec[4], 4);
      const uint8_t l_read_name = rec[8], mapq = rec[9];
      memcpy(&n_cigar, &rec[12], 2);
      memcpy(&flag, &rec[14], 2);
      // reference length = sum of M/D/N/=/X op lengths (ops 0,2,3,7,8)
      int64_t ref_len = 0;
      const size_t co = 32 + l_read_name;
      if (co + 4u * n_cigar > rec.size()) throw std::runtime_error("corrupt BAM cigar");
      for (uint16_t k = 0; k < n_cigar; ++k) {
        uint32_t c;
        memcpy(&c, &rec[co + 4u * k], 4);
        const uint32_t op = c & 0xF;
        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += c >> 4;
      }
      if (cfg_.filter.active) {
        // SemiLazyRecord::intersects (exon-bam/src/indexed_async_batch_stream.rs:66-87)
        if (ref_id < 0 || pos < 0) continue;
        const int64_t s = (int64_t)pos + 1, e = s + ref_len - 1;
        const Region& rg = cfg_.filter.region;
        if (!(ref_id == region_ref_id_ && s <= rg.end && rg.start <= e)) continue;
      }
      b.append((int32_t)flag, ref_id, pos, mapq, ref_len);
      b.append_text(rec.data(), rec.size());
    }
    if (b.is_empty()) return false;
    b.try_into_record_batch(out);
    return true;
  }

  void schema(struct ArrowSchema* out) const {
    std::vector<struct ArrowSchema*> kids = {new_field("i", "flag", false), new_field("C", "mapping_quality", true),
                                             new_field("i", "reference", true, new_field("u", "", false)), new_field("l", "start", true),
                                             new_field("l", "end", true)};
    if (cfg_.projection & 1) kids.push_back(new_field("u", "name", true));
    if (cfg_.projection & 2) kids.push_back(new_field("u", "cigar", false));
    if (cfg_.projection & 4) kids.push_back(new_field("u", "sequence", false));
    if (cfg_.projection & 8) kids.push_back(new_list_field("l", "quality_score"));
    make_schema(out, "+s", "", false, kids);
  }

  std::string header_text;
  std::vector<std::string> ref_names;
  std::vector<int32_t> ref_lengths;
  int n_chunks = -1;
  std::vector<Chunk> planned_chunks;  // the chunks of an indexed scan (the GPU decode path ships exactly these blocks)
  int32_t region_ref_id() const { return region_ref_id_; }  // header index of the pushed-down region's reference (-2: unknown)

 private:
  int32_t read_i32() {
    uint8_t b[4];
    if (!r_->read_exact(b, 4)) throw std::runtime_error("truncated BAM file");
    int32_t v;
    memcpy(&v, b, 4);
    return v;
  }
  std::unique_ptr<RecordSource> r_;
  BAMConfig cfg_;
  int32_t region_ref_id_ = -2;
};

// ======================================================================================================
// SAM (text).  Same schema / device layout as BAM (exon-sam/src/schema_builder.rs:371-402 is shared by
// SAM, BAM and CRAM; text reader: exon-sam/src/batch_reader.rs, array_builder.rs).
// ======================================================================================================
class SAMBatchReader {
 public:
  SAMBatchReader(const std::string& path, Compression c, BAMConfig cfg) : r_(path, c), cfg_(std::move(cfg)) {
    std::string line;
    uint64_t before = r_.consumed();
    while (r_.read_line(&line)) {
      if (line.empty() || line[0] != '@') {
        pending_ = line;
        has_pending_ = !line.empty();
        data_offset_ = (int64_t)before;
        break;
      }
      before = r_.consumed();
      if (line.rfind("@SQ", 0) == 0) {  // @SQ SN:<name> LN:<len>
        std::string name;
        int32_t len = 0;
        size_t start = 0;
        for (size_t i = 0; i <= line.size(); ++i)
          if (i == line.size() || line[i] == '\t') {
            if (line.compare(start, 3, "SN:") == 0) name = line.substr(start + 3, i - start - 3);
            if (line.compare(start, 3, "LN:") == 0) len = atoi(line.c_str() + start + 3);
            start = i + 1;
          }
        ref_names.push_back(name);
        ref_lengths.push_back(len);
      }
      header_text += line + "\n";
    }
    if (cfg_.filter.active) {
      region_ref_id_ = -2;
      for (size_t i = 0; i < ref_names.size(); ++i)
        if (ref_names[i] == cfg_.filter.region.name) region_ref_id_ = (int32_t)i;
    }
  }

  const BAMConfig& config() const { return cfg_; }
  // Uncompressed offset of the first alignment line (header length), for callers that re-open the file themselves
  // (GPU-side BGZF inflate of a bgzip-compressed SAM); the end of the input when there are no alignment lines.
  int64_t data_offset() const { return data_offset_ >= 0 ? data_offset_ : (int64_t)r_.consumed(); }
  // the alignment lines as a raw byte stream (GPU-side parsing); only valid before the first read_batch
  std::unique_ptr<ByteSource> take_stream(std::string* carry) {
    *carry = has_pending_ ? pending_ + "\n" : std::string();
    has_pending_ = false;
    *carry += r_.take_buffered();
    return r_.release_source();
  }

  bool read_batch(struct ArrowArray* out) {
    BAMArrayBuilder b(&ref_names, cfg_.projection);
    std::string line;
    while ((int64_t)b.len() < cfg_.batch_size) {
      if (has_pending_) {
        line.swap(pending_);
        has_pending_ = false;
      } else if (!r_.read_line(&line)) {
        break;
      }
      if (line.empty() || line[0] == '@') continue;
      const char* f[11];
      size_t fl[11];
      int nf = 0;
      size_t start = 0;
      const int want = cfg_.projection ? 11 : 6;
      for (size_t i = 0; i <= line.size() && nf < want; ++i)
        if (i == line.size() || line[i] == '\t') {
          f[nf] = line.data() + start;
          fl[nf] = i - start;
          ++nf;
          start = i + 1;
        }
      if (nf < 6) throw std::runtime_error("SAM record has fewer than 6 fields");
      const int32_t flag = atoi(std::string(f[1], fl[1]).c_str());
      int32_t ref_id = -1;
      if (!(fl[2] == 1 && f[2][0] == '*'))
        for (size_t i = 0; i < ref_names.size(); ++i)
          if (ref_names[i].size() == fl[2] && memcmp(ref_names[i].data(), f[2], fl[2]) == 0) ref_id = (int32_t)i;
      const int64_t pos1 = atoll(std::string(f[3], fl[3]).c_str());  // 1-based, 0 = unavailable
      const int mapq = atoi(std::string(f[4], fl[4]).c_str());
      int64_t ref_len = 0, num = 0;
      if (!(fl[5] == 1 && f[5][0] == '*'))
        for (size_t i = 0; i < fl[5]; ++i) {
          const char ch = f[5][i];
          if (ch >= '0' && ch <= '9') num = num * 10 + (ch - '0');
          else {
            if (ch == 'M' || ch == 'D' || ch == 'N' || ch == '=' || ch == 'X') ref_len += num;
            num = 0;
          }
        }
      if (cfg_.filter.active) {
        if (ref_id < 0 || pos1 < 1) continue;
        const int64_t e = pos1 + ref_len - 1;
        const Region& rg = cfg_.filter.region;
        if (!(ref_id == region_ref_id_ && pos1 <= rg.end && rg.start <= e)) continue;
      }
      if (cfg_.projection) {
        if (nf < 11) throw std::runtime_error("SAM record has fewer than 11 fields");
        b.append_sam_text(f[0], fl[0], f[5], fl[5], f[9], fl[9], f[10], fl[10]);
      }
      b.append(flag, ref_id, pos1 - 1, mapq, ref_len);
    }
    if (b.is_empty()) return false;
    b.try_into_record_batch(out);
    return true;
  }
  void schema(struct ArrowSchema* out) const {
    std::vector<struct ArrowSchema*> kids = {new_field("i", "flag", false), new_field("C", "mapping_quality", true),
                                             new_field("i", "reference", true, new_field("u", "", false)), new_field("l", "start", true),
                                             new_field("l", "end", true)};
    if (cfg_.projection & 1) kids.push_back(new_field("u", "name", true));
    if (cfg_.projection & 2) kids.push_back(new_field("u", "cigar", false));
    if (cfg_.projection & 4) kids.push_back(new_field("u", "sequence", false));
    if (cfg_.projection & 8) kids.push_back(new_list_field("l", "quality_score"));
    make_schema(out, "+s", "", false, kids);
  }
  std::string header_text;
  std::vector<std::string> ref_names;
  std::vector<int32_t> ref_lengths;

 private:
  BufReader r_;
  BAMConfig cfg_;
  std::string pending_;
  bool has_pending_ = false;
  int64_t data_offset_ = -1;
  int32_t region_ref_id_ = -2;
};

// ======================================================================================================
// FASTQ   (name, description?, sequence, quality_scores : exon-fastq/src/config.rs:79-88)
// ======================================================================================================
struct FASTQConfig {
  int64_t batch_size = DEFAULT_BATCH_SIZE;  // sequential reader; the parallel reader emits one batch per ~4 MiB slab
  int threads = 0;
  bool defer_decode = false;  // the text will be taken with take_stream (GPU-side record splitting)
};

class FASTQArrayBuilder : public ExonArrayBuilder {
 public:
  // array_builder.rs:68-102: name up to the first space, description = rest (NULL if empty)
  void append(const std::string& head, const std::string& seq, const std::string& qual) {
    const size_t sp = head.find(' ');
    if (sp == std::string::npos) {
      name_.append_value(head);
      desc_.append_null();
    } else {
      name_.append_value(head.data(), sp);
      if (sp + 1 < head.size()) desc_.append_value(head.data() + sp + 1, head.size() - sp - 1);
      else desc_.append_null();
    }
    seq_.append_value(seq);
    qual_.append_value(qual);
  }
  size_t len() const override { return name_.len(); }
  std::vector<struct ArrowArray*> finish() override { return {name_.finish(), desc_.finish(), seq_.finish(), qual_.finish()}; }

 private:
  Utf8Builder name_, desc_, seq_, qual_;
};

struct FASTQSlab : TextSlab {
  FASTQArrayBuilder b;
  std::vector<struct ArrowArray*> cols;  // finished Arrow columns of the slab (name, description, sequence, quality)
  size_t rows = 0;
};
inline void parse_fastq_slab(FASTQSlab& s, const void*) {
  const char* p = s.data();
  const char* end = p + s.len;
  std::string head, seq, qual;
  auto line = [&](const char** b, size_t* n) {
    if (p >= end) return false;
    const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
    *b = p;
    *n = nl ? (size_t)(nl - p) : (size_t)(end - p);
    if (*n && (*b)[*n - 1] == '\r') --*n;
    p = nl ? nl + 1 : end;
    return true;
  };
  const char *h, *sq, *pl, *ql;
  size_t hn, sn, pn, qn;
  while (line(&h, &hn)) {
    if (hn == 0) continue;
    if (h[0] != '@') throw std::runtime_error("FASTQ record does not start with '@'");
    if (!line(&sq, &sn) || !line(&pl, &pn) || !line(&ql, &qn)) throw std::runtime_error("truncated FASTQ record");
    if (pn == 0 || pl[0] != '+') throw std::runtime_error("FASTQ separator line missing");
    head.assign(h + 1, hn - 1);
    seq.assign(sq, sn);
    qual.assign(ql, qn);
    s.b.append(head, seq, qual);
  }
  s.rows = s.b.len();
}

class FASTQBatchReader {
 public:
  FASTQBatchReader(const std::string& path, Compression c, FASTQConfig cfg) : r_(open_source(path, c, cfg.threads)), cfg_(cfg) {
    const int threads = cfg_.threads > 0 ? cfg_.threads : decode_threads();
    const long fsize = file_size(path);
    // whole 4-line records per slab; blank lines between records are not supported in the parallel path, so it is
    // only taken for inputs that start directly with a record
    if (threads > 1 && fsize >= (8 << 20) && !cfg_.defer_decode)
      pipe_.reset(new SlabPipeline<FASTQSlab>(r_.release_source(), r_.take_buffered(), 4, threads,
                                              [](FASTQSlab& s, const void* c2) { parse_fastq_slab(s, c2); }, nullptr));
  }
  bool read_batch(struct ArrowArray* out) {
    if (pipe_) {
      // a slab is emitted as ONE batch (its Utf8 columns are already contiguous); slabs are ~4 MiB of text
      for (;;) {
        std::unique_ptr<FASTQSlab> s = pipe_->next();
        if (!s) return false;
        if (s->rows == 0) continue;
        s->b.try_into_record_batch(out);
        return true;
      }
    }
    FASTQArrayBuilder b;
    std::string head, seq, plus, qual;
    while ((int64_t)b.len() < cfg_.batch_size) {
      if (!r_.read_line(&head)) break;
      if (head.empty()) continue;
      if (head[0] != '@') throw std::runtime_error("FASTQ record does not start with '@'");
      if (!r_.read_line(&seq) || !r_.read_line(&plus) || !r_.read_line(&qual)) throw std::runtime_error("truncated FASTQ record");
      if (plus.empty() || plus[0] != '+') throw std::runtime_error("FASTQ separator line missing");
      b.append(head.substr(1), seq, qual);
    }
    if (b.is_empty()) return false;
    b.try_into_record_batch(out);
    return true;
  }
  // the whole input as a raw byte stream (GPU-side record splitting); only valid before the first read_batch
  std::unique_ptr<ByteSource> take_stream(std::string* carry) {
    if (pipe_) return nullptr;
    *carry = r_.take_buffered();
    return r_.release_source();
  }
  const FASTQConfig& config() const { return cfg_; }
  void schema(struct ArrowSchema* out) const {
    make_schema(out, "+s", "", false,
                {new_field("u", "name", false), new_field("u", "description", true), new_field("u", "sequence", false),
                 new_field("u", "quality_scores", false)});
  }

 private:
  BufReader r_;
  FASTQConfig cfg_;
  std::unique_ptr<SlabPipeline<FASTQSlab>> pipe_;
};

// ======================================================================================================
// FASTA   (id, description?, sequence : exon-fasta/src/config.rs:164-170)
// ======================================================================================================
struct FASTAConfig {
  int64_t batch_size = DEFAULT_BATCH_SIZE;
};

class FASTAArrayBuilder : public ExonArrayBuilder {
 public:
  // array_builder.rs:114-132: id up to the first whitespace, description = rest (NULL if none)
  void append(const std::string& def, const std::string& seq) {
    size_t ws = 0;
    while (ws < def.size() && def[ws] != ' ' && def[ws] != '\t') ++ws;
    id_.append_value(def.data(), ws);
    size_t d = ws;
    while (d < def.size() && (def[d] == ' ' || def[d] == '\t')) ++d;
    if (d < def.size()) desc_.append_value(def.data() + d, def.size() - d);
    else desc_.append_null();
    seq_.append_value(seq);
  }
  size_t len() const override { return id_.len(); }
  std::vector<struct ArrowArray*> finish() override { return {id_.finish(), desc_.finish(), seq_.finish()}; }

 private:
  Utf8Builder id_, desc_, seq_;
};

class FASTABatchReader {
 public:
  FASTABatchReader(const std::string& path, Compression c, FASTAConfig cfg) : r_(path, c), cfg_(cfg) {}
  // exon-fasta/src/batch_reader.rs:72-99
  bool read_batch(struct ArrowArray* out) {
    FASTAArrayBuilder b;
    std::string line;
    while ((int64_t)b.len() < cfg_.batch_size) {
      if (!have_def_) {
        bool got = false;
        while (r_.read_line(&line)) {
          if (!line.empty() && line[0] == '>') {
            def_ = line.substr(1);
            got = true;
            break;
          }
        }
        if (!got) break;
        have_def_ = true;
      }
      std::string seq;
      bool next_def = false;
      while (r_.read_line(&line)) {
        if (!line.empty() && line[0] == '>') {
          next_def = true;
          break;
        }
        seq += line;
      }
      b.append(def_, seq);
      if (next_def) {
        def_ = line.substr(1);
      } else {
        have_def_ = false;
        break;
      }
    }
    if (b.is_empty()) return false;
    b.try_into_record_batch(out);
    return true;
  }
  void schema(struct ArrowSchema* out) const {
    make_schema(out, "+s", "", false, {new_field("u", "id", false), new_field("u", "description", true), new_field("u", "sequence", false)});
  }

 private:
  BufReader r_;
  FASTAConfig cfg_;
  std::string def_;
  bool have_def_ = false;
};

}  // namespace exon
