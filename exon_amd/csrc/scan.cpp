// scan.cpp -- C ABI of the native decoders (host/formats.h): exon_hip_scan_*.
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "host/bcf.h"
#include "host/formats.h"
#include "internal.h"

struct exon_hip_vcf_parser;  // gpu_parse.hip

struct exon_hip_scan {
  int format = 0;
  bool gpu_parse = false;
  std::string path;
  exon_hip_scan_options opt{};
  std::string info_field_s, region_s;
  exon_hip_vcf_parser* parser = nullptr;  // created by the first GPU-parsed consume; owns the FILTER dictionary
  exon_hip_ctx* parser_ctx = nullptr;
  exon_hip_fastq_parser* fq_parser = nullptr;
  exon::Dictionary gpu_filter_dict;       // names fetched from the parser after the consume
  std::unique_ptr<exon::VCFBatchReader> vcf;
  std::unique_ptr<exon::BAMBatchReader> bam;
  std::unique_ptr<exon::SAMBatchReader> sam;
  std::unique_ptr<exon::BCFBatchReader> bcf;
  std::unique_ptr<exon::FASTQBatchReader> fastq;
  std::unique_ptr<exon::FASTABatchReader> fasta;
  int64_t rows = 0;
  exon::Dictionary bam_dict_view;  // reference names as a dictionary (ids = header order)
};

int exon_hip_stream_push_raw(exon_hip_stream* st, const exon::RawBatch& rb);  // stream.cpp
int exon_hip_stream_launch_scan_columns(exon_hip_stream* st, const exon_hip_column* scan_cols, int n_scan_cols, int64_t n);
int exon_hip_stream_launch_views(exon_hip_stream* st, const uint8_t* d_text, const exon_hip_fastq_views& v);
void* exon_hip_stream_hip_stream(exon_hip_stream* st);
exon_hip_ctx* exon_hip_stream_ctx(exon_hip_stream* st);
int exon_hip_stream_state_copy(exon_hip_stream* st, void* d_snapshot, bool restore);
size_t exon_hip_stream_state_bytes(exon_hip_stream* st);

static exon::Dictionary* dict_of(exon_hip_scan* s, int col) {
  if (s->format == EXON_HIP_FORMAT_BCF && col == 0) return &s->bcf->chrom_dict;
  if (s->format == EXON_HIP_FORMAT_BCF && col == 3) return &s->bcf->filter_dict;
  if (s->format == EXON_HIP_FORMAT_VCF && col == 0) return &s->vcf->chrom_dict;
  if (s->format == EXON_HIP_FORMAT_VCF && col == 3) return s->parser ? &s->gpu_filter_dict : &s->vcf->filter_dict;
  if ((s->format == EXON_HIP_FORMAT_BAM || s->format == EXON_HIP_FORMAT_SAM) && col == 2) return &s->bam_dict_view;
  return nullptr;
}

extern "C" {

int exon_hip_scan_open(const char* path, const exon_hip_scan_options* o, exon_hip_scan** out) {
  if (!path || !o || !out) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_scan_open: NULL argument");
  *out = nullptr;
  try {
    std::unique_ptr<exon_hip_scan> s(new exon_hip_scan());
    s->format = o->format;
    s->path = path;
    s->opt = *o;
    s->info_field_s = o->info_field ? o->info_field : "";
    s->region_s = o->region ? o->region : "";
    const exon::Compression c = o->compression == EXON_HIP_COMPRESSION_GZIP   ? exon::Compression::Gzip
                                : o->compression == EXON_HIP_COMPRESSION_NONE ? exon::Compression::None
                                                                              : exon::Compression::Auto;
    const int64_t bs = o->batch_size > 0 ? o->batch_size : exon::DEFAULT_BATCH_SIZE;
    exon::RegionFilter rf;
    if (o->region && o->region[0]) {
      std::string err;
      if (!exon::parse_region(o->region, &rf.region, &err)) return fail(nullptr, EXON_HIP_EINVAL, "invalid region '%s': %s", o->region, err.c_str());
      rf.active = true;
      rf.use_index = o->use_index != 0;
    }
    switch (o->format) {
      case EXON_HIP_FORMAT_VCF: {
        exon::VCFConfig cfg;
        cfg.batch_size = bs;
        cfg.info_field = o->info_field ? o->info_field : "";
        cfg.filter = rf;
        s->gpu_parse = o->gpu_parse != 0 && !rf.active;  // a pushed-down region filter stays on the host decoder
        cfg.defer_decode = s->gpu_parse;
        s->vcf.reset(new exon::VCFBatchReader(path, c, cfg));
        break;
      }
      case EXON_HIP_FORMAT_BAM: {
        exon::BAMConfig cfg;
        cfg.batch_size = bs;
        cfg.filter = rf;
        s->bam.reset(new exon::BAMBatchReader(path, cfg));
        s->bam_dict_view.names = s->bam->ref_names;
        break;
      }
      case EXON_HIP_FORMAT_BCF: {
        exon::VCFConfig cfg;
        cfg.batch_size = bs;
        cfg.info_field = o->info_field ? o->info_field : "";
        cfg.filter = rf;
        cfg.filter.use_index = false;
        s->bcf.reset(new exon::BCFBatchReader(path, cfg));
        break;
      }
      case EXON_HIP_FORMAT_SAM: {
        exon::BAMConfig cfg;
        cfg.batch_size = bs;
        cfg.filter = rf;
        cfg.filter.use_index = false;
        s->sam.reset(new exon::SAMBatchReader(path, c, cfg));
        s->bam_dict_view.names = s->sam->ref_names;
        break;
      }
      case EXON_HIP_FORMAT_FASTQ: {
        exon::FASTQConfig cfg;
        cfg.batch_size = bs;
        s->gpu_parse = o->gpu_parse != 0;
        cfg.defer_decode = s->gpu_parse;
        s->fastq.reset(new exon::FASTQBatchReader(path, c, cfg));
        break;
      }
      case EXON_HIP_FORMAT_FASTA: {
        exon::FASTAConfig cfg;
        cfg.batch_size = bs;
        s->fasta.reset(new exon::FASTABatchReader(path, c, cfg));
        break;
      }
      default:
        return fail(nullptr, EXON_HIP_EINVAL, "unknown format %d", o->format);
    }
    *out = s.release();
    return EXON_HIP_OK;
  } catch (const std::exception& e) {
    return fail(nullptr, EXON_HIP_EINVAL, "%s", e.what());
  }
}

int exon_hip_scan_schema(exon_hip_scan* s, struct ArrowSchema* out) {
  if (!s || !out) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_scan_schema: NULL argument");
  try {
    if (s->vcf) s->vcf->schema(out);
    else if (s->bam) s->bam->schema(out);
    else if (s->sam) s->sam->schema(out);
    else if (s->bcf) s->bcf->schema(out);
    else if (s->fastq) s->fastq->schema(out);
    else s->fasta->schema(out);
    return EXON_HIP_OK;
  } catch (const std::exception& e) {
    return fail(nullptr, EXON_HIP_EINVAL, "%s", e.what());
  }
}

int exon_hip_scan_next(exon_hip_scan* s, struct ArrowArray* out) {
  if (!s || !out) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_scan_next: NULL argument");
  if (s->gpu_parse) return fail(nullptr, EXON_HIP_ESTATE, "this scan was opened with gpu_parse: use exon_hip_stream_consume_scan");
  try {
    memset(out, 0, sizeof *out);
    bool got;
    if (s->vcf) got = s->vcf->read_batch(out);
    else if (s->bam) got = s->bam->read_batch(out);
    else if (s->sam) got = s->sam->read_batch(out);
    else if (s->bcf) got = s->bcf->read_batch(out);
    else if (s->fastq) got = s->fastq->read_batch(out);
    else got = s->fasta->read_batch(out);
    if (!got) return 1;
    s->rows += out->length;
    return EXON_HIP_OK;
  } catch (const std::exception& e) {
    return fail(nullptr, EXON_HIP_EINVAL, "%s", e.what());
  }
}

int exon_hip_scan_dictionary_size(exon_hip_scan* s, int32_t column, int32_t* size) {
  if (!s || !size) return fail(nullptr, EXON_HIP_EINVAL, "NULL argument");
  exon::Dictionary* d = dict_of(s, column);
  if (!d) return fail(nullptr, EXON_HIP_EINVAL, "column %d is not dictionary-encoded", column);
  *size = (int32_t)d->names.size();
  return EXON_HIP_OK;
}

int exon_hip_scan_dictionary_intern(exon_hip_scan* s, int32_t column, const char* name, int32_t* id) {
  if (!s || !name || !id) return fail(nullptr, EXON_HIP_EINVAL, "NULL argument");
  exon::Dictionary* d = dict_of(s, column);
  if (!d) return fail(nullptr, EXON_HIP_EINVAL, "column %d is not dictionary-encoded", column);
  if (s->format == EXON_HIP_FORMAT_BAM || s->format == EXON_HIP_FORMAT_SAM) {  // reference ids are fixed by the header
    *id = d->find(name);
    return EXON_HIP_OK;
  }
  *id = d->lookup_or_insert(name, strlen(name));
  return EXON_HIP_OK;
}

int exon_hip_scan_dictionary_value(exon_hip_scan* s, int32_t column, int32_t id, const char** name) {
  if (!s || !name) return fail(nullptr, EXON_HIP_EINVAL, "NULL argument");
  exon::Dictionary* d = dict_of(s, column);
  if (!d || id < 0 || id >= (int32_t)d->names.size()) return fail(nullptr, EXON_HIP_EINVAL, "no dictionary entry %d in column %d", id, column);
  *name = d->names[(size_t)id].c_str();
  return EXON_HIP_OK;
}

int exon_hip_scan_rows(exon_hip_scan* s, int64_t* rows) {
  if (!s || !rows) return fail(nullptr, EXON_HIP_EINVAL, "NULL argument");
  *rows = s->rows;
  return EXON_HIP_OK;
}

int exon_hip_scan_index_chunks(exon_hip_scan* s, int32_t* n) {
  if (!s || !n) return fail(nullptr, EXON_HIP_EINVAL, "NULL argument");
  *n = s->vcf ? s->vcf->n_chunks : s->bam ? s->bam->n_chunks : -1;
  return EXON_HIP_OK;
}

int exon_hip_index_query(const char* index_path, int32_t is_bai, const char* ref_name, int32_t ref_id, int64_t start,
                         int64_t end, uint64_t* starts, uint64_t* ends, int32_t cap, int32_t* n_chunks) {
  if (!index_path || !n_chunks) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_index_query: NULL argument");
  try {
    const exon::BinningIndex idx = is_bai ? exon::read_bai(index_path) : exon::read_tabix(index_path);
    int id = ref_id;
    if (!is_bai && ref_name) {
      id = -1;
      for (size_t i = 0; i < idx.names.size(); ++i)
        if (idx.names[i] == ref_name) id = (int)i;
    }
    const std::vector<exon::Chunk> chunks = exon::query_index(idx, id, start, end);
    *n_chunks = (int32_t)chunks.size();
    for (int32_t i = 0; i < *n_chunks && i < cap; ++i) {
      if (starts) starts[i] = chunks[(size_t)i].start;
      if (ends) ends[i] = chunks[(size_t)i].end;
    }
    return EXON_HIP_OK;
  } catch (const std::exception& e) {
    return fail(nullptr, EXON_HIP_EINVAL, "%s", e.what());
  }
}

int exon_hip_scan_close(exon_hip_scan* s) {
  if (s && s->parser) exon_hip_vcf_parser_destroy(s->parser);
  if (s && s->fq_parser) exon_hip_fastq_parser_destroy(s->fq_parser);
  delete s;
  return EXON_HIP_OK;
}

}  // extern "C"

// plain files are read with positional reads from several threads (one fread stream tops out near 6 GB/s);
// compressed inputs go through the (block-parallel) inflating source
struct SlabReader {
  explicit SlabReader(exon::ByteSource* s) : src(s) { plain = src->plain_file(&fd, &foff); }
  size_t read(uint8_t* dst, size_t n) {
    if (!plain) {
      size_t have = 0;
      while (have < n) {
        const size_t got = src->read(dst + have, n - have);
        if (got == 0) break;
        have += got;
      }
      return have;
    }
    const int T = 8;
    const size_t per = (n + T - 1) / T;
    size_t got[T] = {0};
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) {
      const size_t o = (size_t)t * per;
      if (o >= n) break;
      const size_t len = std::min(per, n - o);
      th.emplace_back([this, &got, dst, t, o, len] {
        size_t done = 0;
        while (done < len) {
          const ssize_t r = pread(fd, dst + o + done, len - done, (off_t)(foff + (int64_t)(o + done)));
          if (r <= 0) break;
          done += (size_t)r;
        }
        got[t] = done;
      });
    }
    for (auto& x : th) x.join();
    size_t total = 0;
    for (int t = 0; t < T; ++t) {
      total += got[t];
      if (got[t] < std::min(per, n > (size_t)t * per ? n - (size_t)t * per : 0)) break;  // EOF inside this chunk
    }
    foff += (int64_t)total;
    return total;
  }
  exon::ByteSource* src;
  bool plain = false;
  int fd = -1;
  int64_t foff = 0;
};

static size_t slab_bytes() {
  size_t slab = 64u << 20;
  if (const char* v = getenv("EXON_HIP_GPU_PARSE_SLAB_MB")) {
    const long mb = atol(v);
    if (mb >= 1 && mb <= 1024) slab = (size_t)mb << 20;
  }
  return slab;
}

// file -> pinned slab -> HBM -> GPU parser -> fused kernel.  A background thread fills the next pinned slab while
// the current one is copied and parsed.  Returns 1 when a slab held rows the device could not decide: the caller
// restores the state and re-decodes the file on the host.
static int consume_vcf_gpu(exon_hip_stream* st, exon_hip_scan* scan, int64_t* rows_out) {
  exon_hip_ctx* ctx = exon_hip_stream_ctx(st);
  hipStream_t hs = (hipStream_t)exon_hip_stream_hip_stream(st);
  std::string carry;
  std::unique_ptr<exon::ByteSource> src = scan->vcf->take_stream(&carry);
  if (!src) return fail(ctx, EXON_HIP_ESTATE, "scan already consumed");
  const size_t slab = slab_bytes();
  if (!scan->parser) {
    std::vector<const char*> names;
    for (const auto& c : scan->vcf->header.contigs) names.push_back(c.c_str());
    int rc = exon_hip_vcf_parser_create(ctx, names.data(), (int32_t)names.size(),
                                        scan->info_field_s.empty() ? nullptr : scan->info_field_s.c_str(),
                                        (int64_t)slab + 65536, &scan->parser);
    if (rc) return rc;
    scan->parser_ctx = ctx;
  }
  uint8_t* h_buf[2] = {nullptr, nullptr};
  uint8_t* d_buf[2] = {nullptr, nullptr};
  auto cleanup = [&]() {
    for (int k = 0; k < 2; ++k) {
      if (h_buf[k]) hipHostFree(h_buf[k]);
      if (d_buf[k]) hipFree(d_buf[k]);
    }
  };
  const size_t cap = slab + 65536;
  for (int k = 0; k < 2; ++k)
    if (hipHostMalloc((void**)&h_buf[k], cap) != hipSuccess || hipMalloc((void**)&d_buf[k], cap + 64) != hipSuccess) {
      cleanup();
      return fail(ctx, EXON_HIP_ENOMEM, "slab buffers (%zu bytes) could not be allocated", cap);
    }
  SlabReader rd(src.get());
  auto read_some = [&](uint8_t* dst, size_t n) -> size_t { return rd.read(dst, n); };
  // reader: fills h_buf[k] with [carry | fresh bytes], cuts at the last newline, keeps the tail as the next carry
  struct Filled { size_t n = 0; bool eof = false; std::exception_ptr err; };
  auto fill = [&](int k, Filled* f) {
    try {
      size_t have = carry.size();
      if (have > cap) throw std::runtime_error("VCF line longer than the slab size");
      memcpy(h_buf[k], carry.data(), have);
      carry.clear();
      for (;;) {
        while (have < slab) {
          const size_t got = read_some(h_buf[k] + have, slab - have);
          if (got == 0) { f->eof = true; break; }
          have += got;
        }
        size_t cut = have;
        if (!f->eof) {
          while (cut > 0 && h_buf[k][cut - 1] != '\n') --cut;
          if (cut == 0) throw std::runtime_error("VCF line longer than the slab size");
        } else if (have > 0 && h_buf[k][have - 1] != '\n') {
          h_buf[k][have++] = '\n';  // last line without a terminator (room: cap > slab)
          cut = have;
        }
        carry.assign(reinterpret_cast<const char*>(h_buf[k]) + cut, have - cut);
        f->n = cut;
        return;
      }
    } catch (...) {
      f->err = std::current_exception();
    }
  };
  int64_t total = 0;
  int rc = EXON_HIP_OK;
  Filled cur, nxt;
  fill(0, &cur);
  int k = 0;
  while (rc == EXON_HIP_OK) {
    if (cur.err) {
      try { std::rethrow_exception(cur.err); } catch (const std::exception& e) { rc = fail(ctx, EXON_HIP_EINVAL, "%s", e.what()); }
      break;
    }
    std::thread reader;
    const bool more = !cur.eof;
    if (more) reader = std::thread([&, k] { fill(k ^ 1, &nxt); });  // overlaps with the copy + parse below
    if (cur.n > 0) {
      hipError_t e = hipMemcpyAsync(d_buf[k], h_buf[k], cur.n, hipMemcpyHostToDevice, hs);
      if (e != hipSuccess) rc = fail(ctx, EXON_HIP_EDEVICE, "H2D of a text slab: %s", hipGetErrorString(e));
      exon_hip_vcf_columns cols;
      if (!rc) rc = exon_hip_vcf_parser_parse(scan->parser, hs, d_buf[k], (int64_t)cur.n, &cols);
      if (!rc && cols.n_undecided > 0) rc = 1;  // host fallback
      if (!rc && cols.n_rows > 0) {
        exon_hip_column sc[5];
        memset(sc, 0, sizeof sc);
        sc[0].values = cols.chrom_id;
        sc[1].values = cols.pos;
        sc[1].validity = cols.pos_valid;
        sc[2].values = cols.qual;
        sc[2].validity = cols.qual_valid;
        sc[3].values = cols.filter_id;
        sc[4].values = cols.info;
        sc[4].validity = cols.info_valid;
        for (auto& c : sc) c.length = cols.n_rows;
        rc = exon_hip_stream_launch_scan_columns(st, sc, 5, cols.n_rows);
        // the parser's column buffers are reused by the next slab: the fused kernel must be done with them first
        if (!rc && hipStreamSynchronize(hs) != hipSuccess) rc = fail(ctx, EXON_HIP_EDEVICE, "stream synchronize failed");
        total += cols.n_rows;
      }
    }
    if (reader.joinable()) reader.join();
    if (!more) break;
    cur = nxt;
    nxt = Filled();
    k ^= 1;
  }
  cleanup();
  if (rc == EXON_HIP_OK) {
    // FILTER dictionary -> scan (names in id order)
    int32_t nf = 0;
    std::vector<char> buf(1 << 20);
    rc = exon_hip_vcf_parser_filters(scan->parser, buf.data(), buf.size(), &nf);
    if (!rc) {
      scan->gpu_filter_dict.names.clear();
      size_t o = 0;
      for (int32_t i = 0; i < nf; ++i) {
        scan->gpu_filter_dict.names.emplace_back(buf.data() + o);
        o += scan->gpu_filter_dict.names.back().size() + 1;
      }
    }
    scan->rows += total;
    if (rows_out) *rows_out = total;
  }
  return rc;
}


// FASTQ: file -> pinned slab -> HBM -> newline index + per-read views -> K5 over the views (no columns are built).
// The device reports where the last whole record of a slab ends; the tail is carried in front of the next slab, which
// the reader thread has meanwhile filled behind a reserved gap.  Returns 1 for "decode on the host instead".
static int consume_fastq_gpu(exon_hip_stream* st, exon_hip_scan* scan, int64_t* rows_out) {
  exon_hip_ctx* ctx = exon_hip_stream_ctx(st);
  hipStream_t hs = (hipStream_t)exon_hip_stream_hip_stream(st);
  std::string carry;
  std::unique_ptr<exon::ByteSource> src = scan->fastq->take_stream(&carry);
  if (!src) return fail(ctx, EXON_HIP_ESTATE, "scan already consumed");
  const size_t slab = slab_bytes();
  const size_t gap = std::max<size_t>(slab / 4, 1u << 20);  // room for the carried tail (< 1 record + 1 line)
  const size_t cap = gap + slab + 64;
  if (!scan->fq_parser) {
    int rc = exon_hip_fastq_parser_create(ctx, (int64_t)cap, &scan->fq_parser);
    if (rc) return rc;
  }
  uint8_t* h_buf[2] = {nullptr, nullptr};
  uint8_t* d_buf[2] = {nullptr, nullptr};
  auto cleanup = [&]() {
    for (int k = 0; k < 2; ++k) {
      if (h_buf[k]) hipHostFree(h_buf[k]);
      if (d_buf[k]) hipFree(d_buf[k]);
    }
  };
  for (int k = 0; k < 2; ++k)
    if (hipHostMalloc((void**)&h_buf[k], cap) != hipSuccess || hipMalloc((void**)&d_buf[k], cap + 64) != hipSuccess) {
      cleanup();
      return fail(ctx, EXON_HIP_ENOMEM, "slab buffers (%zu bytes) could not be allocated", cap);
    }
  SlabReader rd(src.get());
  struct Filled { size_t n = 0; bool eof = false; std::exception_ptr err; };
  auto fill = [&](int k, Filled* f) {  // fresh bytes behind the gap
    try {
      f->n = rd.read(h_buf[k] + gap, slab);
      f->eof = f->n < slab;
    } catch (...) {
      f->err = std::current_exception();
    }
  };
  int64_t total = 0;
  int rc = EXON_HIP_OK;
  Filled cur, nxt;
  fill(0, &cur);
  int k = 0;
  while (rc == EXON_HIP_OK) {
    if (cur.err) {
      try { std::rethrow_exception(cur.err); } catch (const std::exception& e) { rc = fail(ctx, EXON_HIP_EINVAL, "%s", e.what()); }
      break;
    }
    if (carry.size() > gap) { rc = 1; break; }  // a record larger than the gap: host decoder
    uint8_t* base = h_buf[k] + gap - carry.size();
    memcpy(base, carry.data(), carry.size());
    size_t n = carry.size() + cur.n;
    carry.clear();
    if (cur.eof && n > 0 && base[n - 1] != '\n') base[n++] = '\n';  // last line without a terminator
    std::thread reader;
    const bool more = !cur.eof;
    if (more) reader = std::thread([&, k] { fill(k ^ 1, &nxt); });  // overlaps with the copy + index + histogram
    if (n > 0) {
      hipError_t e = hipMemcpyAsync(d_buf[k], base, n, hipMemcpyHostToDevice, hs);
      if (e != hipSuccess) rc = fail(ctx, EXON_HIP_EDEVICE, "H2D of a text slab: %s", hipGetErrorString(e));
      exon_hip_fastq_views v;
      if (!rc) rc = exon_hip_fastq_parser_parse(scan->fq_parser, hs, d_buf[k], (int64_t)n, more ? 0 : 1, &v);
      if (!rc && v.n_undecided > 0) rc = 1;
      if (!rc && more && v.consumed_bytes == 0) rc = 1;  // not one whole record in a slab
      if (!rc) {
        carry.assign(reinterpret_cast<const char*>(base) + v.consumed_bytes, n - (size_t)v.consumed_bytes);
        if (!more && !carry.empty()) rc = 1;
      }
      if (!rc && v.n_reads > 0) {
        rc = exon_hip_stream_launch_views(st, d_buf[k], v);  // asynchronous: overlaps with preparing the next slab
        total += v.n_reads;
      }
    }
    if (reader.joinable()) reader.join();
    if (!more) break;
    cur = nxt;
    nxt = Filled();
    k ^= 1;
  }
  if (hipStreamSynchronize(hs) != hipSuccess && rc == EXON_HIP_OK) rc = fail(ctx, EXON_HIP_EDEVICE, "stream synchronize failed");
  cleanup();
  if (rc == EXON_HIP_OK) {
    scan->rows += total;
    if (rows_out) *rows_out = total;
  }
  return rc;
}

extern "C" {

int exon_hip_stream_consume_scan(exon_hip_stream* st, exon_hip_scan* scan, int64_t* rows) {
  if (!st || !scan) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_stream_consume_scan: NULL argument");
  int64_t n = 0;
  if (scan->gpu_parse && scan->fastq) {
    exon_hip_ctx* ctx = exon_hip_stream_ctx(st);
    void* snap = nullptr;
    const size_t sb = exon_hip_stream_state_bytes(st);
    if (hipMalloc(&snap, sb ? sb : 16) != hipSuccess) return fail(ctx, EXON_HIP_ENOMEM, "state snapshot allocation failed");
    int rc = exon_hip_stream_state_copy(st, snap, false);
    if (!rc) rc = consume_fastq_gpu(st, scan, rows);
    if (rc == 1) {  // restore and re-decode on the host
      rc = exon_hip_stream_state_copy(st, snap, true);
      hipStreamSynchronize((hipStream_t)exon_hip_stream_hip_stream(st));
      hipFree(snap);
      if (rc) return rc;
      scan->gpu_parse = false;
      try {
        exon::FASTQConfig cfg = scan->fastq->config();
        cfg.defer_decode = false;
        const exon::Compression c = scan->opt.compression == EXON_HIP_COMPRESSION_GZIP   ? exon::Compression::Gzip
                                    : scan->opt.compression == EXON_HIP_COMPRESSION_NONE ? exon::Compression::None
                                                                                         : exon::Compression::Auto;
        scan->fastq.reset(new exon::FASTQBatchReader(scan->path, c, cfg));
      } catch (const std::exception& e) {
        return fail(nullptr, EXON_HIP_EINVAL, "%s", e.what());
      }
    } else {
      hipFree(snap);
      return rc;
    }
  }
  if (scan->gpu_parse && scan->vcf) {
    // speculative GPU decode; on undecidable rows restore the state and fall back to the host decoder
    exon_hip_ctx* ctx = exon_hip_stream_ctx(st);
    void* snap = nullptr;
    const size_t sb = exon_hip_stream_state_bytes(st);
    if (hipMalloc(&snap, sb ? sb : 16) != hipSuccess) return fail(ctx, EXON_HIP_ENOMEM, "state snapshot allocation failed");
    int rc = exon_hip_stream_state_copy(st, snap, false);
    if (!rc) rc = consume_vcf_gpu(st, scan, rows);
    if (rc == 1) {
      rc = exon_hip_stream_state_copy(st, snap, true);
      hipStreamSynchronize((hipStream_t)exon_hip_stream_hip_stream(st));
      hipFree(snap);
      if (rc) return rc;
      if (scan->parser) {
        exon_hip_vcf_parser_destroy(scan->parser);
        scan->parser = nullptr;
      }
      scan->gpu_parse = false;
      try {
        exon::VCFConfig cfg = scan->vcf->config();
        cfg.defer_decode = false;
        const exon::Compression c = scan->opt.compression == EXON_HIP_COMPRESSION_GZIP   ? exon::Compression::Gzip
                                    : scan->opt.compression == EXON_HIP_COMPRESSION_NONE ? exon::Compression::None
                                                                                         : exon::Compression::Auto;
        scan->vcf.reset(new exon::VCFBatchReader(scan->path, c, cfg));
      } catch (const std::exception& e) {
        return fail(nullptr, EXON_HIP_EINVAL, "%s", e.what());
      }
      // fall through to the host paths below
    } else {
      hipFree(snap);
      return rc;
    }
  }
  // fast path: a multi-threaded VCF scan hands its slabs over as raw vectors (no Arrow batch in between)
  if (scan->vcf) {
    try {
      exon::RawBatch rb;
      bool end = false;
      while (scan->vcf->read_raw(&rb, &end)) {
        const int rc = exon_hip_stream_push_raw(st, rb);
        if (rc < 0) return rc;
        n += rb.rows;
        scan->rows += rb.rows;
      }
      if (end) {
        if (rows) *rows = n;
        return EXON_HIP_OK;
      }
    } catch (const std::exception& e) {
      return fail(nullptr, EXON_HIP_EINVAL, "%s", e.what());
    }
  }
  for (;;) {
    struct ArrowArray batch;
    int rc = exon_hip_scan_next(scan, &batch);
    if (rc == 1) break;
    if (rc < 0) return rc;
    n += batch.length;
    rc = exon_hip_stream_push(st, &batch);  // moves the batch
    if (rc < 0) return rc;
  }
  if (rows) *rows = n;
  return EXON_HIP_OK;
}

}  // extern "C"
