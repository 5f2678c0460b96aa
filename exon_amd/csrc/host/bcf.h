// bcf.h -- BCF2 (binary VCF) record decoder emitting the VCF device layout.
//
// Reference: exon-bcf (config / batch reader / array builder, same Arrow schema as VCF) behind
// exon-core/src/datasources/bcf/; pinned by exon-core/src/session_context/exon_context_ext.rs:1053-1090
// (index.bcf: 621 records, 191 on chromosome "1").  Record layout: VCF 4.x specification section 6 (BCF2): BGZF stream,
// "BCF\2\2", text header, then per record `l_shared, l_indiv`, fixed fields (CHROM id, 0-based POS, rlen, QUAL with the
// 0x7F800001 missing sentinel, n_info | n_allele << 16, n_fmt << 24 | n_sample), typed ID / alleles / FILTER (dictionary
// indexes) / INFO (dictionary index -> typed value).
#pragma once
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "formats.h"

namespace exon {

class BCFBatchReader {
 public:
  BCFBatchReader(const std::string& path, VCFConfig cfg) : cfg_(std::move(cfg)) {
    r_.reset(new StreamSource(path, Compression::Gzip, cfg_.threads));
    uint8_t magic[5];
    if (!r_->read_exact(magic, 5) || memcmp(magic, "BCF\2\2", 5) != 0) throw std::runtime_error("not a BCF2 file: " + path);
    uint8_t lb[4];
    if (!r_->read_exact(lb, 4)) throw std::runtime_error("truncated BCF header");
    uint32_t l_text;
    memcpy(&l_text, lb, 4);
    std::string text((size_t)l_text, '\0');
    if (l_text && !r_->read_exact(reinterpret_cast<uint8_t*>(&text[0]), l_text)) throw std::runtime_error("truncated BCF header");
    // dictionaries: contigs by ##contig order (or IDX), strings (FILTER/INFO/FORMAT ids) with PASS = 0
    std::map<std::string, int> str_idx;
    auto add_string = [&](const std::string& id, const std::string& idx_attr) {
      if (str_idx.count(id)) return;
      const int idx = idx_attr.empty() ? (int)strings_.size() : atoi(idx_attr.c_str());
      if ((int)strings_.size() <= idx) strings_.resize((size_t)idx + 1);
      strings_[(size_t)idx] = id;
      str_idx[id] = idx;
    };
    add_string("PASS", "0");
    size_t start = 0;
    while (start < text.size()) {
      size_t nl = text.find('\n', start);
      if (nl == std::string::npos) nl = text.size();
      const std::string line = text.substr(start, nl - start);
      start = nl + 1;
      if (line.rfind("##contig=", 0) == 0) {
        const std::string id = header_attr(line, "ID"), idx = header_attr(line, "IDX");
        const int k = idx.empty() ? (int)header.contigs.size() : atoi(idx.c_str());
        if ((int)header.contigs.size() <= k) header.contigs.resize((size_t)k + 1);
        header.contigs[(size_t)k] = id;
      } else if (line.rfind("##FILTER=", 0) == 0) {
        header.filters.push_back(header_attr(line, "ID"));
        add_string(header_attr(line, "ID"), header_attr(line, "IDX"));
      } else if (line.rfind("##INFO=", 0) == 0) {
        header.infos.emplace_back(header_attr(line, "ID"), header_attr(line, "Number") + "|" + header_attr(line, "Type"));
        add_string(header_attr(line, "ID"), header_attr(line, "IDX"));
      } else if (line.rfind("##FORMAT=", 0) == 0) {
        add_string(header_attr(line, "ID"), header_attr(line, "IDX"));
      } else if (line.rfind("#CHROM", 0) == 0) {
        size_t col = 0, st = 0;
        for (size_t i = 0; i <= line.size(); ++i)
          if (i == line.size() || line[i] == '\t') {
            if (col >= 9) header.samples.push_back(line.substr(st, i - st));
            ++col;
            st = i + 1;
          }
      }
    }
    for (const auto& c : header.contigs) chrom_dict.names.push_back(c);
    // typed INFO fields of the scan (same rules as the VCF reader: formats.h InfoSpec); keys are header-string indexes here
    info_specs = resolve_info_specs(cfg_.info_field, header.infos);
    info_dicts.assign(info_specs.size(), Dictionary());
    info_keys_.clear();
    for (const auto& sp : info_specs) {
      auto it = str_idx.find(sp.name);
      if (it == str_idx.end()) throw std::runtime_error("INFO field " + sp.name + " is not declared in the header");
      info_keys_.push_back(it->second);
    }
    info_key_ = info_keys_.empty() ? -1 : info_keys_[0];
    if (cfg_.filter.active) {
      region_chrom_ = -2;
      for (size_t i = 0; i < header.contigs.size(); ++i)
        if (header.contigs[i] == cfg_.filter.region.name) region_chrom_ = (int)i;
    }
  }

  bool read_batch(struct ArrowArray* out) {
    PrimitiveBuilder<int32_t> chrom, filter;
    PrimitiveBuilder<int64_t> pos;
    PrimitiveBuilder<float> qual;
    const size_t K = info_specs.size();
    std::vector<PrimitiveBuilder<float>> info_f(K);
    std::vector<PrimitiveBuilder<int32_t>> info_i(K);
    std::vector<ListBuilder<float>> info_lf(K);    // 'F'
    std::vector<ListBuilder<int32_t>> info_li(K);  // 'I' values, 'S' dictionary ids
    // projected text columns: BCF records go through the reference's EAGER builder (exon-bcf/src/batch_reader.rs:72 ->
    // exon-vcf/src/array_builder/eager_array_builder.rs:112-134): ids and alternate bases are lists WITH their items and are
    // never NULL (an empty list when there is none) -- unlike the lazy builder VCF text takes
    ListUtf8Builder ids, alts;
    Utf8Builder refs;
    std::vector<uint8_t> rec;
    size_t rows = 0;
    auto typed_string = [&](size_t* o, size_t end, const char** p, size_t* n) {
      int type, count;
      typed_header(rec, o, end, &type, &count);
      if (type != 7 && !(type == 0 && count == 0)) throw std::runtime_error("corrupt BCF record: a string was expected");
      if (*o + (size_t)count > end) throw std::runtime_error("corrupt BCF typed value");
      *p = reinterpret_cast<const char*>(rec.data() + *o);
      *n = type == 7 ? (size_t)count : 0;
      *o += *n;
    };
    while ((int64_t)rows < cfg_.batch_size) {
      uint8_t lens[8];
      if (!r_->read_exact(lens, 8)) break;
      uint32_t l_shared, l_indiv;
      memcpy(&l_shared, lens, 4);
      memcpy(&l_indiv, lens + 4, 4);
      if (l_shared < 24) throw std::runtime_error("corrupt BCF record");
      rec.resize((size_t)l_shared + l_indiv);
      if (!r_->read_exact(rec.data(), rec.size())) throw std::runtime_error("truncated BCF record");
      int32_t chrom_id, pos0;
      uint32_t qbits, nia;
      memcpy(&chrom_id, &rec[0], 4);
      memcpy(&pos0, &rec[4], 4);
      memcpy(&qbits, &rec[12], 4);
      memcpy(&nia, &rec[16], 4);
      const int n_info = (int)(nia & 0xFFFF), n_allele = (int)(nia >> 16);
      if (cfg_.filter.active) {  // same per-record interval hit as the VCF reader
        const Region& rg = cfg_.filter.region;
        const int64_t p1 = (int64_t)pos0 + 1;
        if (chrom_id != region_chrom_ || p1 < rg.start || p1 > rg.end) continue;
      }
      size_t o = 24;
      const size_t end = l_shared;
      if (!cfg_.projection) {
        skip_typed(rec, &o, end);                               // ID
        for (int a = 0; a < n_allele; ++a) skip_typed(rec, &o, end);  // REF + ALTs
      } else {
        const char* tp;
        size_t tn;
        typed_string(&o, end, &tp, &tn);  // ID: ';'-separated, "." = none
        if ((cfg_.projection & 1) && !(tn == 0 || (tn == 1 && tp[0] == '.'))) {
          size_t a = 0;
          for (size_t i = 0; i <= tn; ++i)
            if (i == tn || tp[i] == ';') {
              ids.items.append_value(tp + a, i - a);
              a = i + 1;
            }
        }
        if (cfg_.projection & 1) ids.close_row();
        for (int a = 0; a < n_allele; ++a) {
          typed_string(&o, end, &tp, &tn);
          if (a == 0) {
            if (cfg_.projection & 2) refs.append_value(tp, tn);
          } else if (cfg_.projection & 4) {
            alts.items.append_value(tp, tn);
          }
        }
        if ((cfg_.projection & 2) && n_allele == 0) refs.append_value("", 0);
        if (cfg_.projection & 4) alts.close_row();
      }
      // FILTER: typed int vector of dictionary indexes; empty = '.'
      std::string fl;
      {
        int type, count;
        typed_header(rec, &o, end, &type, &count);
        for (int i = 0; i < count; ++i) {
          const int64_t v = read_int(rec, &o, end, type);
          if (v < 0 || v >= (int64_t)strings_.size()) throw std::runtime_error("BCF FILTER index out of range");
          if (i) fl += ';';
          fl += strings_[(size_t)v];
        }
      }
      // INFO: (typed key, typed value) pairs
      bool have[MAX_INFO_FIELDS] = {};
      float fv[MAX_INFO_FIELDS] = {};
      int32_t sv[MAX_INFO_FIELDS] = {};
      for (int k = 0; k < n_info; ++k) {
        int kt, kc;
        typed_header(rec, &o, end, &kt, &kc);
        const int64_t key = kc ? read_int(rec, &o, end, kt) : -1;
        int vt, vc;
        typed_header(rec, &o, end, &vt, &vc);
        const size_t vbytes = (size_t)vc * type_size(vt);
        if (o + vbytes > end) throw std::runtime_error("corrupt BCF INFO");
        for (size_t q = 0; q < K; ++q) {
          if (key != info_keys_[q] || have[q]) continue;
          const char kind = info_specs[q].kind;
          if (kind == 'b') {
            have[q] = true;  // a Flag is true by being there (typed value: missing type, or an int8 1)
          } else if (kind == 'i' && vc >= 1) {
            // Type=Integer: int8 / int16 / int32 typed value, widened to Int32 exactly; the type's minimum is 'missing'
            if (vt >= 1 && vt <= 3) {
              size_t oo = o;
              const int64_t v = read_int(rec, &oo, end, vt);
              const int64_t missing = vt == 1 ? -128 : vt == 2 ? -32768 : (int64_t)INT32_MIN;
              if (v != missing) {
                sv[q] = (int32_t)v;
                have[q] = true;
              }
            }
          } else if (kind == 'f' && vc >= 1) {
            if (vt == 5) {
              uint32_t b;
              memcpy(&b, &rec[o], 4);
              if (b != 0x7F800001u && b != 0x7F800002u) {
                memcpy(&fv[q], &b, 4);
                have[q] = true;
              }
            } else if (vt >= 1 && vt <= 3) {
              size_t oo = o;
              const int64_t v = read_int(rec, &oo, end, vt);
              const int64_t missing = vt == 1 ? -128 : vt == 2 ? -32768 : (int64_t)INT32_MIN;
              if (v != missing) {
                fv[q] = (float)v;
                have[q] = true;
              }
            }
          } else if (info_kind_is_list(kind) && vc >= 1) {
            // typed vector -> List<item>: the type's 'missing' value is a NULL item, its 'end of vector' value ends the list
            // (VCF specification 6.3.3); a character vector holds the ','-joined text of a String list
            if (kind == 'S' && vt == 7) {
              size_t len = (size_t)vc;
              while (len > 0 && rec[o + len - 1] == 0) --len;
              if (!(len == 1 && rec[o] == '.') && len > 0) {
                size_t a = 0;
                while (a <= len) {
                  size_t e = a;
                  while (e < len && rec[o + e] != ',') ++e;
                  if (e == a || (e - a == 1 && rec[o + a] == '.')) info_li[q].items.append_null(0);
                  else info_li[q].items.append_value(info_dicts[q].lookup_or_insert(reinterpret_cast<const char*>(&rec[o + a]), e - a));
                  a = e + 1;
                }
                info_li[q].close_row();
                have[q] = true;
              }
            } else if (kind == 'F' && vt == 5) {
              int items = 0;
              if (vc == 1) {  // a single 'missing' item is `key=.`: the whole value is missing
                uint32_t b0;
                memcpy(&b0, &rec[o], 4);
                if (b0 == 0x7F800001u) continue;
              }
              for (int e = 0; e < vc; ++e) {
                uint32_t b;
                memcpy(&b, &rec[o + 4 * (size_t)e], 4);
                if (b == 0x7F800002u) break;
                float f;
                memcpy(&f, &b, 4);
                if (b == 0x7F800001u) info_lf[q].items.append_null(0.f);
                else info_lf[q].items.append_value(f);
                ++items;
              }
              if (items > 0) {
                info_lf[q].close_row();
                have[q] = true;
              }
            } else if ((kind == 'I' || kind == 'F') && vt >= 1 && vt <= 3) {
              size_t oo = o;
              const int64_t missing = vt == 1 ? -128 : vt == 2 ? -32768 : (int64_t)INT32_MIN;
              int items = 0;
              if (vc == 1) {  // a single 'missing' item is `key=.`: the whole value is missing
                size_t o1 = o;
                if (read_int(rec, &o1, end, vt) == missing) continue;
              }
              for (int e = 0; e < vc; ++e) {
                const int64_t v = read_int(rec, &oo, end, vt);
                if (v == missing + 1) break;  // end of vector
                if (kind == 'I') {
                  if (v == missing) info_li[q].items.append_null(0);
                  else info_li[q].items.append_value((int32_t)v);
                } else {
                  if (v == missing) info_lf[q].items.append_null(0.f);
                  else info_lf[q].items.append_value((float)v);
                }
                ++items;
              }
              if (items > 0) {
                if (kind == 'I') info_li[q].close_row();
                else info_lf[q].close_row();
                have[q] = true;
              }
            }
          } else if (kind == 's' && vt == 7 && vc >= 1) {
            size_t len = (size_t)vc;
            while (len > 0 && rec[o + len - 1] == 0) --len;  // strings may be NUL-padded
            if (!(len == 1 && rec[o] == '.') && len > 0) {
              sv[q] = info_dicts[q].lookup_or_insert(reinterpret_cast<const char*>(&rec[o]), len);
              have[q] = true;
            }
          }
        }
        o += vbytes;
      }
      if (chrom_id < 0 || chrom_id >= (int)chrom_dict.names.size()) throw std::runtime_error("BCF CHROM index out of range");
      chrom.append_value(chrom_id);
      if (pos0 >= 0) pos.append_value((int64_t)pos0 + 1);
      else pos.append_null(0);  // POS 0 (the telomere) has no variant_start: NULL, as in the VCF path
      if (qbits == 0x7F800001u) qual.append_null(0.f);
      else {
        float q;
        memcpy(&q, &qbits, 4);
        qual.append_value(q);
      }
      filter.append_value(filter_dict.lookup_or_insert(fl.data(), fl.size()));
      for (size_t q = 0; q < K; ++q) {
        if (info_specs[q].kind == 'F') {
          if (!have[q]) info_lf[q].append_null();  // (a present list closed its row when it was decoded)
        } else if (info_specs[q].kind == 'I' || info_specs[q].kind == 'S') {
          if (!have[q]) info_li[q].append_null();
        } else if (info_specs[q].kind == 'f') {
          if (have[q]) info_f[q].append_value(fv[q]);
          else info_f[q].append_null(0.f);
        } else {
          if (have[q]) info_i[q].append_value(info_specs[q].kind == 'b' ? 1 : sv[q]);
          else info_i[q].append_null(0);
        }
      }
      ++rows;
    }
    if (rows == 0) return false;
    std::vector<struct ArrowArray*> kids = {chrom.finish(utf8_array(chrom_dict.names)), pos.finish(), qual.finish(),
                                            filter.finish(utf8_array(filter_dict.names))};
    for (size_t q = 0; q < K; ++q) {
      if (info_specs[q].kind == 'f') kids.push_back(info_f[q].finish());
      else if (info_specs[q].kind == 'i') kids.push_back(info_i[q].finish());
      else if (info_specs[q].kind == 'F') kids.push_back(info_lf[q].finish());
      else if (info_specs[q].kind == 'I') kids.push_back(info_li[q].finish());
      else if (info_specs[q].kind == 'S') kids.push_back(info_li[q].finish(utf8_array(info_dicts[q].names)));
      else if (info_specs[q].kind == 's') kids.push_back(info_i[q].finish(utf8_array(info_dicts[q].names)));
      else {
        struct ArrowArray* a = static_cast<struct ArrowArray*>(malloc(sizeof *a));
        make_boolean(a, info_i[q].valid, info_i[q].valid);
        kids.push_back(a);
      }
    }
    if (cfg_.projection & 1) kids.push_back(ids.finish());
    if (cfg_.projection & 2) kids.push_back(refs.finish());
    if (cfg_.projection & 4) kids.push_back(alts.finish());
    make_struct(out, (int64_t)rows, std::move(kids));
    return true;
  }

  void schema(struct ArrowSchema* out) const {
    std::vector<struct ArrowSchema*> kids = {new_field("i", "chrom", false, new_field("u", "", false)), new_field("l", "pos", true),
                                             new_field("f", "qual", true), new_field("i", "filter", false, new_field("u", "", false))};
    for (const auto& sp : info_specs) {
      const std::string name = "info." + sp.name;
      if (sp.kind == 'f') kids.push_back(new_field("f", name.c_str(), true));
      else if (sp.kind == 'i') kids.push_back(new_field("i", name.c_str(), true));
      else if (sp.kind == 'F') kids.push_back(new_list_field("f", name.c_str()));
      else if (sp.kind == 'I') kids.push_back(new_list_field("i", name.c_str()));
      else if (sp.kind == 'S') kids.push_back(new_list_field("i", name.c_str(), new_field("u", "", false)));
      else if (sp.kind == 'b') kids.push_back(new_field("b", name.c_str(), true));
      else kids.push_back(new_field("i", name.c_str(), true, new_field("u", "", false)));
    }
    if (cfg_.projection & 1) kids.push_back(new_list_field("u", "id"));
    if (cfg_.projection & 2) kids.push_back(new_field("u", "ref", false));
    if (cfg_.projection & 4) kids.push_back(new_list_field("u", "alt"));
    make_schema(out, "+s", "", false, kids);
  }

  // what a GPU-side decoder of the same file needs (bcf_parse.hip)
  const VCFConfig& config() const { return cfg_; }
  const std::vector<std::string>& strings() const { return strings_; }
  int info_key() const { return info_key_; }
  const std::vector<int>& info_keys() const { return info_keys_; }
  int64_t data_offset() const { return (int64_t)static_cast<StreamSource*>(r_.get())->r.consumed(); }  // header bytes

  VCFHeader header;
  Dictionary chrom_dict, filter_dict;
  std::vector<InfoSpec> info_specs;
  std::vector<Dictionary> info_dicts;

 private:
  std::vector<int> info_keys_;
  static int type_size(int t) { return t == 1 ? 1 : t == 2 ? 2 : t == 3 ? 4 : t == 5 ? 4 : t == 7 ? 1 : 0; }
  static int64_t read_int(const std::vector<uint8_t>& b, size_t* o, size_t end, int type) {
    const int sz = type_size(type);
    if (type < 1 || type > 3 || *o + (size_t)sz > end) throw std::runtime_error("corrupt BCF typed integer");
    int64_t v;
    if (type == 1) v = (int8_t)b[*o];
    else if (type == 2) {
      int16_t x;
      memcpy(&x, &b[*o], 2);
      v = x;
    } else {
      int32_t x;
      memcpy(&x, &b[*o], 4);
      v = x;
    }
    *o += (size_t)sz;
    return v;
  }
  static void typed_header(const std::vector<uint8_t>& b, size_t* o, size_t end, int* type, int* count) {
    if (*o >= end) throw std::runtime_error("corrupt BCF typed value");
    const uint8_t d = b[(*o)++];
    *type = d & 0xF;
    *count = d >> 4;
    if (*count == 15) {  // the real count follows as a typed integer
      int ct, cc;
      typed_header(b, o, end, &ct, &cc);
      *count = (int)read_int(b, o, end, ct);
    }
  }
  static void skip_typed(const std::vector<uint8_t>& b, size_t* o, size_t end) {
    int t, c;
    typed_header(b, o, end, &t, &c);
    *o += (size_t)c * type_size(t);
    if (*o > end) throw std::runtime_error("corrupt BCF typed value");
  }

  std::unique_ptr<RecordSource> r_;
  VCFConfig cfg_;
  std::vector<std::string> strings_;
  int info_key_ = -1, region_chrom_ = -2;
};

}  // namespace exon
