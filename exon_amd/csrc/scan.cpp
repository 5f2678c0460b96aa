// scan.cpp -- C ABI of the native decoders (host/formats.h): exon_hip_scan_*.
#include <pthread.h>
#include <sched.h>
#include <time.h>
#include <unistd.h>

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <map>
#include <deque>
#include <array>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "host/bcf.h"
#include "host/cram.h"
#include "host/formats.h"
#include "host/parallel.h"
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
  exon_hip_bam_parser* bam_parser = nullptr;
  exon_hip_bcf_parser* bcf_parser = nullptr;
  exon_hip_sam_parser* sam_parser = nullptr;
  struct GpuExporter* exporter = nullptr;  // exon_hip_scan_bind_ctx: batches (exon_hip_scan_next) come out of the GPU pipeline
  // opened with gpu_parse but with INFO keys only the host reader builds (String / Character values, list-valued keys): batches
  // (exon_hip_scan_next) come from the host reader, but a consume_scan whose plan reads none of those columns still takes the
  // GPU pipeline -- the device parser decodes and validates the list keys and does not look at the string keys at all
  bool gpu_candidate = false;
  bool gpu_inflated = false;  // the last GPU-parsed consume also inflated BGZF blocks on the device
  bool gpu_decoded = false;   // the last consume decoded every record on the device (no host fallback)
  exon::Dictionary gpu_filter_dict;       // names fetched from the parser after the consume
  std::unique_ptr<exon::VCFBatchReader> vcf;
  std::unique_ptr<exon::BAMBatchReader> bam;
  std::unique_ptr<exon::SAMBatchReader> sam;
  std::unique_ptr<exon::CRAMBatchReader> cram;
  std::unique_ptr<exon::BCFBatchReader> bcf;
  std::unique_ptr<exon::FASTQBatchReader> fastq;
  std::unique_ptr<exon::FASTABatchReader> fasta;
  int64_t rows = 0;
  exon::Dictionary bam_dict_view;  // reference names as a dictionary (ids = header order)
  // pushed-down region filter (vcf_region_filter / bam_region_filter) on the GPU decode path
  exon::RegionFilter region;
  uint8_t* d_region_mask = nullptr;         // row mask of the slab being consumed (grown on demand)
  size_t region_mask_cap = 0;
  unsigned long long* d_region_pass = nullptr;  // rows kept so far in this consume
  exon_hip_ctx* region_ctx = nullptr;
  ExonTextScratch* text_scratch = nullptr;  // device buffers of the projected string / list columns (text_columns.hip)
};

// Batches from the GPU decode pipeline (exon_hip_scan_bind_ctx + exon_hip_scan_next on a scan opened with gpu_parse): a producer
// thread drives the same slab pipeline exon_hip_stream_consume_scan drives -- file bytes to HBM, BGZF inflate, record parse,
// region mask, all on the device -- but instead of a fused filter + aggregate kernel every slab's columns come back over PCIe
// and are cut into batch_size-row Arrow batches, built by the SAME column builders the host readers use (so the layout is
// theirs: dictionary-encoded chrom / filter / reference, NULL rules included).  The consumer takes them off a bounded queue
// (the producer waits when it is full).  If the device cannot decide a record, the producer goes on with the host reader
// from the first row it has not emitted.  For queries whose plan the optimizer rule does not match: the reference's
// <Fmt>Scan::execute (exon-core/src/datasources/vcf/scanner.rs:142-162, bam/scanner.rs:138-158) is exactly this surface.
struct GpuExporter {
  exon_hip_ctx* ctx = nullptr;
  exon_hip_plan* plan = nullptr;
  exon_hip_stream* st = nullptr;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv_put, cv_get;
  std::deque<struct ArrowArray*> q;
  size_t cap = 64;
  bool started = false, done = false, stop = false;
  int rc = 0;
  std::string err;
  int64_t emitted = 0;                       // rows handed to the queue so far
  std::vector<std::string> final_filters;    // the FILTER dictionary when the producer has finished
  std::vector<std::vector<std::string>> final_info_names;  // ... and the String INFO keys' dictionaries
  bool decoded_on_gpu = false, inflated_on_gpu = false;
  // the host reader that takes over when the device hands the file back.  It lives HERE while the producer thread runs: the
  // scan's own reader (which exon_hip_scan_schema / _dictionary_* read from the consumer's thread) is never touched by the
  // producer; gpu_next moves the fallback into the scan after the thread has been joined.
  std::unique_ptr<exon::VCFBatchReader> fb_vcf;
  std::unique_ptr<exon::BAMBatchReader> fb_bam;
  std::unique_ptr<exon::SAMBatchReader> fb_sam;
  std::unique_ptr<exon::BCFBatchReader> fb_bcf;
  std::unique_ptr<exon::FASTQBatchReader> fb_fastq;
  bool handed_over = false;
  // A slab's columns cross PCIe while the NEXT slab is inflated and parsed: export_slab copies what it needs device to device
  // into a staging slot (the parsers reuse their output buffers), a copy stream takes it to the pinned block, and the slab's
  // batches are cut when the next slab arrives (or the scan ends) -- `pending` waits for the copy and emits them.
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done[2] = {nullptr, nullptr};
  uint8_t* d_stage[2] = {nullptr, nullptr};
  size_t stage_cap[2] = {0, 0};
  uint64_t n_exports = 0;
  std::function<int()> pending;
};

// copies of one slab: device source -> staging slot (on the pipeline's stream) -> host destination (on the copy stream)
struct SlabCopier {
  GpuExporter* ex;
  hipStream_t hs;
  int slot;
  size_t used = 0;
  struct Item {
    void* dst;
    size_t off, bytes;
  };
  std::vector<Item> items;
  hipError_t err = hipSuccess;
  SlabCopier(GpuExporter* e, hipStream_t s) : ex(e), hs(s), slot((int)(e->n_exports & 1)) {}
  // room for `bytes` more (called before the first add of a group of copies whose sizes are known)
  bool reserve(size_t bytes) {
    const size_t need = used + bytes + 4096;
    if (ex->stage_cap[slot] >= need) return true;
    if (used) return false;  // (never: every caller reserves its whole need up front)
    if (ex->d_stage[slot]) hipFree(ex->d_stage[slot]);
    ex->d_stage[slot] = nullptr;
    ex->stage_cap[slot] = 0;
    const size_t cap = need + need / 4;
    if (hipMalloc((void**)&ex->d_stage[slot], cap) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    ex->stage_cap[slot] = cap;
    return true;
  }
  void add(void* dst, const void* src, size_t bytes) {
    if (!bytes || err != hipSuccess) return;
    const size_t off = (used + 255) & ~size_t(255);
    if (off + bytes > ex->stage_cap[slot]) {
      err = hipErrorOutOfMemory;
      return;
    }
    err = hipMemcpyAsync(ex->d_stage[slot] + off, src, bytes, hipMemcpyDeviceToDevice, hs);
    items.push_back(Item{dst, off, bytes});
    used = off + bytes;
  }
  // the staged bytes start for the host; ev_done[slot] fires when they have arrived
  hipError_t launch() {
    if (err != hipSuccess) return err;
    hipError_t e = hipEventRecord(ex->ev_ready, hs);
    if (e == hipSuccess) e = hipStreamWaitEvent(ex->copy_stream, ex->ev_ready, 0);
    for (const Item& it : items)
      if (e == hipSuccess) e = hipMemcpyAsync(it.dst, ex->d_stage[slot] + it.off, it.bytes, hipMemcpyDeviceToHost, ex->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(ex->ev_done[slot], ex->copy_stream);
    return e;
  }
};

static int gpu_next(exon_hip_scan* s, struct ArrowArray* out);
static void gpu_export_shutdown(exon_hip_scan* s);
int exon_hip_stream_push_raw(exon_hip_stream* st, const exon::RawBatch& rb);  // stream.cpp
int exon_hip_stream_launch_scan_columns(exon_hip_stream* st, const exon_hip_column* scan_cols, int n_scan_cols, int64_t n,
                                        const uint8_t* row_mask);
int exon_hip_stream_plan_first_column(exon_hip_stream* st);
int exon_hip_stream_plan_kind(exon_hip_stream* st);
int exon_hip_stream_plan_column(exon_hip_stream* st, int arg);
void exon_hip_stream_set_value_types(exon_hip_stream* st, int x_type, int y_type);
int exon_hip_stream_set_null_group(exon_hip_stream* st, std::function<int32_t()> id_of_null);
int exon_hip_stream_launch_views(exon_hip_stream* st, const uint8_t* d_text, const exon_hip_fastq_views& v);
void* exon_hip_stream_hip_stream(exon_hip_stream* st);
exon_hip_ctx* exon_hip_stream_ctx(exon_hip_stream* st);
int exon_hip_stream_state_copy(exon_hip_stream* st, void* d_snapshot, bool restore, int64_t* rows_pushed);
size_t exon_hip_stream_state_bytes(exon_hip_stream* st);
bool exon_hip_stream_is_keyed(exon_hip_stream* st);
int exon_hip_stream_begin_scan(exon_hip_stream* st, bool* tracked, bool* redirected);
int exon_hip_stream_end_scan(exon_hip_stream* st, const std::vector<std::string>* scan_keys, bool tracked, bool redirected, bool ok);
bool exon_hip_stream_region_contig(exon_hip_stream* st, std::string* name);
int exon_hip_stream_set_region_id(exon_hip_stream* st, int32_t id);

// BGZF inputs of GPU-parsed scans are inflated on the GPU too (EXON_HIP_GPU_INFLATE=0: host threads inflate)
static bool gpu_inflate_enabled() {
  const char* v = getenv("EXON_HIP_GPU_INFLATE");
  return !(v && v[0] == '0');
}
static bool gpu_gzip_enabled() {
  const char* v = getenv("EXON_HIP_GPU_GZIP");
  return !(v && v[0] == '0');
}
// gzip magic + deflate method, and not BGZF
static bool is_plain_gzip(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  unsigned char h[3] = {0, 0, 0};
  const bool ok = fread(h, 1, 3, f) == 3 && h[0] == 0x1f && h[1] == 0x8b && h[2] == 8;
  fclose(f);
  return ok && !exon::BgzfParallelSource::is_bgzf(path);
}
static bool wants_gpu_inflate(const exon_hip_scan_options* o, const char* path) {
  return o->gpu_parse != 0 && gpu_inflate_enabled() && o->compression != EXON_HIP_COMPRESSION_NONE && exon::BgzfParallelSource::is_bgzf(path);
}

// File-level codecs the reference's openers accept through `file_compression_type.convert_stream`
// (exon-core/src/datasources/fastq/file_opener.rs:93-105) that this library does not read: zstd, bzip2, xz.  Told apart
// by their magic numbers (none of the formats' own first bytes -- '#', '@', '>', "CRAM", the gzip magic -- collide).
static const char* unsupported_codec(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) return nullptr;  // the reader reports the missing file
  unsigned char m[6] = {0, 0, 0, 0, 0, 0};
  const size_t got = fread(m, 1, 6, f);
  fclose(f);
  if (got >= 4 && m[0] == 0x28 && m[1] == 0xB5 && m[2] == 0x2F && m[3] == 0xFD) return "zstd";
  if (got >= 4 && m[0] == 'B' && m[1] == 'Z' && m[2] == 'h' && m[3] >= '1' && m[3] <= '9') return "bzip2";
  if (got >= 6 && m[0] == 0xFD && m[1] == '7' && m[2] == 'z' && m[3] == 'X' && m[4] == 'Z' && m[5] == 0) return "xz";
  return nullptr;
}

// INFO kinds the GPU pipeline of THIS scan decodes: Float / Integer / Flag everywhere; Number=1 String / Character (dictionary ids,
// the dictionary built on the device) for VCF text
static bool info_kind_decoded_on_gpu(const exon_hip_scan* s, char kind) { return exon::info_kind_on_device(kind) || (kind == 's' && s->vcf != nullptr); }

static exon::Dictionary* dict_of(exon_hip_scan* s, int col) {
  if (s->format == EXON_HIP_FORMAT_BCF && col == 0) return &s->bcf->chrom_dict;
  if (s->format == EXON_HIP_FORMAT_BCF && col == 3) return s->bcf_parser ? &s->gpu_filter_dict : &s->bcf->filter_dict;
  if (s->format == EXON_HIP_FORMAT_VCF && col == 0) return &s->vcf->chrom_dict;
  if (s->format == EXON_HIP_FORMAT_VCF && col == 3) return s->parser ? &s->gpu_filter_dict : &s->vcf->filter_dict;
  if ((s->format == EXON_HIP_FORMAT_BAM || s->format == EXON_HIP_FORMAT_SAM || s->format == EXON_HIP_FORMAT_CRAM) && col == 2) return &s->bam_dict_view;
  // string INFO fields (scan columns 4 ..) are dictionary-encoded by the host readers
  if (s->format == EXON_HIP_FORMAT_VCF && col >= 4 && (size_t)(col - 4) < s->vcf->info_specs.size() && (s->vcf->info_specs[(size_t)(col - 4)].kind == 's' || s->vcf->info_specs[(size_t)(col - 4)].kind == 'S'))
    return &s->vcf->info_dicts[(size_t)(col - 4)];
  if (s->format == EXON_HIP_FORMAT_BCF && col >= 4 && (size_t)(col - 4) < s->bcf->info_specs.size() && (s->bcf->info_specs[(size_t)(col - 4)].kind == 's' || s->bcf->info_specs[(size_t)(col - 4)].kind == 'S'))
    return &s->bcf->info_dicts[(size_t)(col - 4)];
  return nullptr;
}

extern "C" {

int exon_hip_scan_open(const char* path, const exon_hip_scan_options* o, exon_hip_scan** out) {
  if (!path || !o || !out) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_scan_open: NULL argument");
  *out = nullptr;
  if (o->compression != EXON_HIP_COMPRESSION_NONE)
    if (const char* codec = unsupported_codec(path))
      return fail(nullptr, EXON_HIP_EUNSUPPORTED, "%s: %s-compressed input (only uncompressed, gzip and BGZF files are read; decompress or re-compress with bgzip)", path, codec);
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
    s->region = rf;
    if (o->projection && o->format != EXON_HIP_FORMAT_VCF && o->format != EXON_HIP_FORMAT_BAM && o->format != EXON_HIP_FORMAT_BCF && o->format != EXON_HIP_FORMAT_SAM)
      return fail(nullptr, EXON_HIP_EUNSUPPORTED, "exon_hip_scan_options.projection: the id / ref / alt (/ info / formats) and name / cigar / sequence / quality_score columns are built for VCF, BCF, BAM and SAM scans");
    switch (o->format) {
      case EXON_HIP_FORMAT_VCF: {
        exon::VCFConfig cfg;
        cfg.batch_size = bs;
        cfg.info_field = o->info_field ? o->info_field : "";
        cfg.filter = rf;
        cfg.projection = o->projection;
        if (o->projection & ~31ull) return fail(nullptr, EXON_HIP_EINVAL, "projection 0x%llx: VCF knows EXON_HIP_PROJECT_VCF_ID / _REF / _ALT / _INFO / _FORMATS", (unsigned long long)o->projection);
        // a pushed-down region filter rides along as a row mask (k_region_mask); with use_index the host plans the
        // tabix chunks and only their BGZF blocks are shipped (indexed scans are BGZF by definition)
        s->gpu_parse = o->gpu_parse != 0 && (!rf.use_index || (rf.active && wants_gpu_inflate(o, path)));
        // info / formats as text are the reference's re-printed entries (host/vcf_text.h: number formatting per header type):
        // the host reader builds them, so such a scan decodes there
        if (o->projection & (EXON_HIP_PROJECT_VCF_INFO | EXON_HIP_PROJECT_VCF_FORMATS)) {
          s->gpu_candidate = s->gpu_parse;  // (a fused plan never reads them: exon_hip_stream_consume_scan may still decode on the device)
          s->gpu_parse = false;
        }
        // EXON_HIP_REFERENCE_QUIRKS=1: an indexed VCF scan reproduces the reference's unfiltered tail after a full batch of
        // hits (exon-vcf/src/indexed_async_batch_stream.rs:143-154) -- a property of its per-chunk record loop, so the host
        // reader runs it; default: every record is tested (what vcf_region_filter documents)
        if (const char* qv = getenv("EXON_HIP_REFERENCE_QUIRKS"); qv && qv[0] == '1' && rf.active && rf.use_index) {
          cfg.reference_tail_quirk = true;
          s->gpu_parse = false;
        }
        cfg.defer_decode = s->gpu_parse;
        if (s->gpu_parse && wants_gpu_inflate(o, path)) cfg.threads = 1;  // only the header is read on the host
        s->vcf.reset(new exon::VCFBatchReader(path, c, cfg));
        if (s->gpu_parse) {
          bool string_info = false;
          for (const auto& sp : s->vcf->info_specs) string_info |= !info_kind_decoded_on_gpu(s.get(), sp.kind);
          if (string_info) {  // string / list INFO columns are built by the host reader only: batches come from there
            s->gpu_parse = false;
            s->gpu_candidate = true;
            cfg.defer_decode = false;
            cfg.threads = 0;
            s->vcf.reset(new exon::VCFBatchReader(path, c, cfg));
          }
        }
        break;
      }
      case EXON_HIP_FORMAT_BAM: {
        exon::BAMConfig cfg;
        cfg.batch_size = bs;
        cfg.filter = rf;
        cfg.projection = o->projection;
        if (o->projection & ~15ull) return fail(nullptr, EXON_HIP_EINVAL, "projection 0x%llx: BAM knows EXON_HIP_PROJECT_BAM_NAME / _CIGAR / _SEQUENCE / _QUALITY_SCORES", (unsigned long long)o->projection);
        // BAM is BGZF by definition: the GPU path inflates and splits records on the device or is not taken at all
        s->gpu_parse = wants_gpu_inflate(o, path);  // with a region: row mask on the device, BAI chunks planned on the host
        if (s->gpu_parse) cfg.threads = 1;  // only the header is read on the host
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
        cfg.projection = o->projection;
        if (o->projection & ~7ull) return fail(nullptr, EXON_HIP_EINVAL, "projection 0x%llx: BCF knows EXON_HIP_PROJECT_VCF_ID / _REF / _ALT", (unsigned long long)o->projection);
        s->gpu_parse = wants_gpu_inflate(o, path);  // BCF is BGZF by definition; a region becomes a row mask (id / ref / alt come from the device too)
        if (s->gpu_parse) cfg.threads = 1;  // only the header is read on the host
        s->bcf.reset(new exon::BCFBatchReader(path, cfg));
        if (s->gpu_parse) {
          bool string_info = false;
          for (const auto& sp : s->bcf->info_specs) string_info |= !exon::info_kind_on_device(sp.kind);
          if (string_info) {
            s->gpu_parse = false;
            s->gpu_candidate = true;
            cfg.threads = 0;
            s->bcf.reset(new exon::BCFBatchReader(path, cfg));
          }
        }
        break;
      }
      case EXON_HIP_FORMAT_SAM: {
        exon::BAMConfig cfg;
        cfg.batch_size = bs;
        cfg.filter = rf;
        cfg.filter.use_index = false;
        cfg.projection = o->projection;
        if (o->projection & ~15ull) return fail(nullptr, EXON_HIP_EINVAL, "projection 0x%llx: SAM knows EXON_HIP_PROJECT_BAM_NAME / _CIGAR / _SEQUENCE / _QUALITY_SCORES", (unsigned long long)o->projection);
        s->gpu_parse = o->gpu_parse != 0;  // (the text columns come from the device too: text_columns.hip, k_sam_measure / k_sam_fill)
        s->sam.reset(new exon::SAMBatchReader(path, c, cfg));
        s->bam_dict_view.names = s->sam->ref_names;
        break;
      }
      case EXON_HIP_FORMAT_CRAM: {  // host decoder only: the columns go to HBM through the staging path
        exon::BAMConfig cfg;
        cfg.batch_size = bs;
        cfg.filter = rf;
        cfg.filter.use_index = false;
        s->gpu_parse = false;
        s->cram.reset(new exon::CRAMBatchReader(path, cfg));
        s->bam_dict_view.names = s->cram->ref_names;
        break;
      }
      case EXON_HIP_FORMAT_FASTQ: {
        exon::FASTQConfig cfg;
        cfg.batch_size = bs;
        s->gpu_parse = o->gpu_parse != 0;
        cfg.defer_decode = s->gpu_parse;
        if (s->gpu_parse && wants_gpu_inflate(o, path)) cfg.threads = 1;
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
    else if (s->cram) s->cram->schema(out);
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
  if (s->gpu_parse && s->exporter) return gpu_next(s, out);
  if (s->gpu_parse)
    return fail(nullptr, EXON_HIP_ESTATE, "this scan was opened with gpu_parse: use exon_hip_stream_consume_scan, or exon_hip_scan_bind_ctx for batches from the GPU pipeline");
  try {
    memset(out, 0, sizeof *out);
    bool got;
    if (s->vcf) got = s->vcf->read_batch(out);
    else if (s->bam) got = s->bam->read_batch(out);
    else if (s->sam) got = s->sam->read_batch(out);
    else if (s->cram) got = s->cram->read_batch(out);
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
  if (s->format == EXON_HIP_FORMAT_BAM || s->format == EXON_HIP_FORMAT_SAM || s->format == EXON_HIP_FORMAT_CRAM) {  // reference ids are fixed by the header
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

int exon_hip_scan_bind_ctx(exon_hip_scan* s, exon_hip_ctx* ctx) {
  if (!s || !ctx) return fail(ctx, EXON_HIP_EINVAL, "exon_hip_scan_bind_ctx: NULL argument");
  if (s->exporter) return fail(ctx, EXON_HIP_ESTATE, "the scan is bound to a context already");
  if (!(s->vcf || s->bcf || s->bam || s->sam || s->fastq))
    return fail(ctx, EXON_HIP_EUNSUPPORTED, "batches from the GPU pipeline: VCF, BCF, BAM, SAM and FASTQ scans (FASTA / CRAM batches come from the host readers)");
  if (!s->gpu_parse)  // not opened with gpu_parse, or String / list-valued INFO keys were named: the host reader builds those columns
    return fail(ctx, EXON_HIP_EUNSUPPORTED, "this scan's batches come from the host reader (opened without gpu_parse, or it names INFO keys only the host reader builds)");
  if (s->rows != 0) return fail(ctx, EXON_HIP_ESTATE, "the scan has been read from already");
  s->exporter = new GpuExporter();
  s->exporter->ctx = ctx;
  return EXON_HIP_OK;
}

int exon_hip_scan_close(exon_hip_scan* s) {
  if (s && s->exporter) gpu_export_shutdown(s);
  if (s && s->parser) exon_hip_vcf_parser_destroy(s->parser);
  if (s && s->fq_parser) exon_hip_fastq_parser_destroy(s->fq_parser);
  if (s && s->bam_parser) exon_hip_bam_parser_destroy(s->bam_parser);
  if (s && s->bcf_parser) exon_hip_bcf_parser_destroy(s->bcf_parser);
  if (s && s->sam_parser) exon_hip_sam_parser_destroy(s->sam_parser);
  if (s && s->text_scratch) exon_text_scratch_destroy(s->text_scratch);
  if (s && s->d_region_mask) hipFree(s->d_region_mask);
  if (s && s->d_region_pass) hipFree(s->d_region_pass);
  delete s;
  return EXON_HIP_OK;
}

}  // extern "C"

// The file pipelines' host threads (the reader and its read pool) run on the CPUs of the GPU's NUMA node: they copy the file out of
// the page cache into the pinned staging ring the DMA engine reads a moment later, and on a two-socket host the same scan takes
// 90 ms with them on the GPU's socket and 115-126 ms on the other one -- left to the scheduler it was one or the other from run to
// run (profiles/r5_reader_numa.log).  The caller's own thread is not touched.  EXON_HIP_READER_AFFINITY=0 turns it off.
static const cpu_set_t* gpu_local_cpus(int device);
const void* exon_hip_gpu_local_cpus(int device) { return gpu_local_cpus(device); }
void exon_hip_run_on(const void* cpus) {
  if (cpus) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), static_cast<const cpu_set_t*>(cpus));
}
static const cpu_set_t* gpu_local_cpus(int device) {
  static std::mutex mu;
  static std::map<int, cpu_set_t*> cache;  // nullptr: unknown / not applicable
  std::lock_guard<std::mutex> g(mu);
  auto it = cache.find(device);
  if (it != cache.end()) return it->second;
  cpu_set_t* set = nullptr;
  const char* sw = getenv("EXON_HIP_READER_AFFINITY");
  char bdf[64] = {0};
  if (!(sw && sw[0] == '0') && hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device) == hipSuccess) {
    for (char* c = bdf; *c; ++c)
      if (*c >= 'A' && *c <= 'Z') *c = (char)(*c - 'A' + 'a');
    int node = -1;
    if (FILE* f = fopen((std::string("/sys/bus/pci/devices/") + bdf + "/numa_node").c_str(), "r")) {
      if (fscanf(f, "%d", &node) != 1) node = -1;
      fclose(f);
    }
    char list[4096] = {0};
    if (node >= 0)
      if (FILE* f = fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r")) {
        if (!fgets(list, (int)sizeof list, f)) list[0] = 0;
        fclose(f);
      }
    cpu_set_t allowed, local;
    CPU_ZERO(&local);
    for (const char* q = list; *q && *q != '\n';) {  // "0-63,128-191"
      char* e = nullptr;
      const long a = strtol(q, &e, 10);
      if (e == q) break;
      long b = a;
      if (*e == '-') b = strtol(e + 1, &e, 10);
      for (long c = a; c <= b && c < CPU_SETSIZE; ++c)
        if (c >= 0) CPU_SET((int)c, &local);
      if (*e != ',') break;
      q = e + 1;
    }
    if (sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
      cpu_set_t both;
      CPU_AND(&both, &allowed, &local);
      if (CPU_COUNT(&both) >= 4 && CPU_COUNT(&both) < CPU_COUNT(&allowed)) set = new cpu_set_t(both);  // (a one-node host: nothing to choose)
    }
  }
  cache[device] = set;
  return set;
}
static void run_on(const cpu_set_t* cpus) {
  if (cpus) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), cpus);
}

// plain files are read with positional reads from several threads (one fread stream tops out near 6 GB/s);
// compressed inputs go through the (block-parallel) inflating source
struct SlabReader {
  explicit SlabReader(exon::ByteSource* s) : src(s) { plain = src->plain_file(&fd, &foff); }
  const cpu_set_t* cpus = nullptr;  // where the pool's threads run (set before the first read)
  ~SlabReader() {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  // Asynchronous form: begin() starts reading the next `n` bytes of the input into `dst` and returns; finish() waits and
  // returns how many arrived (fewer = end of input).  One read may be outstanding.  Positional files: a pool of threads that
  // lives as long as the reader (spawning eight threads per 8 MB piece cost more than the reads themselves) copies slices
  // out of the page cache; 8 threads saturate it (16 and 24 measured no faster: the copy out of the page cache competes with
  // the DMA engine for the same memory).  EXON_HIP_READ_THREADS overrides.  Other sources are read at finish().
  void begin(uint8_t* dst, size_t n) {
    pend_dst_ = dst;
    pend_n_ = n;
    begun_ = true;
    if (!plain || n == 0) return;
    static const int T = [] {
      int t = 8;
      if (const char* v = getenv("EXON_HIP_READ_THREADS")) {
        const int x = atoi(v);
        if (x >= 1 && x <= 32) t = x;
      }
      return t;
    }();
    if (th_.empty())
      for (int t = 0; t < T; ++t) th_.emplace_back([this, t] { run(t); });
    const size_t per = std::max<size_t>((n + T - 1) / T, 256u << 10);
    std::lock_guard<std::mutex> g(mu_);
    njobs_ = 0;
    for (size_t o = 0; o < n; o += per) jobs_[njobs_++] = Job{dst + o, foff + (int64_t)o, std::min(per, n - o), 0};
    pending_ = njobs_;
    ++gen_;
    cv_.notify_all();
  }
  size_t finish() {
    if (!begun_) return 0;
    begun_ = false;
    if (!plain) {
      size_t have = 0;
      while (have < pend_n_) {
        const size_t got = src->read(pend_dst_ + have, pend_n_ - have);
        if (got == 0) break;
        have += got;
      }
      return have;
    }
    if (pend_n_ == 0) return 0;
    std::unique_lock<std::mutex> lk(mu_);
    cv_done_.wait(lk, [this] { return pending_ == 0; });
    size_t total = 0;
    for (int j = 0; j < njobs_; ++j) {
      total += jobs_[j].got;
      if (jobs_[j].got < jobs_[j].len) break;  // EOF inside this slice
    }
    foff += (int64_t)total;
    return total;
  }
  size_t read(uint8_t* dst, size_t n) {
    begin(dst, n);
    return finish();
  }
  exon::ByteSource* src;
  bool plain = false;
  int fd = -1;
  int64_t foff = 0;

 private:
  struct Job {
    uint8_t* dst;
    int64_t off;
    size_t len, got;
  };
  void run(int t) {
    run_on(cpus);
    uint64_t seen = 0;
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        if (t >= njobs_) continue;
        j = jobs_[t];
      }
      size_t done = 0;
      while (done < j.len) {
        const ssize_t r = pread(fd, j.dst + done, j.len - done, (off_t)(j.off + (int64_t)done));
        if (r <= 0) break;
        done += (size_t)r;
      }
      std::lock_guard<std::mutex> g(mu_);
      jobs_[t].got = done;
      if (--pending_ == 0) cv_done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, cv_done_;
  std::vector<std::thread> th_;
  Job jobs_[32];
  int njobs_ = 0, pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false, begun_ = false;
  uint8_t* pend_dst_ = nullptr;
  size_t pend_n_ = 0;
};

static size_t slab_bytes(bool bgzf_parsed_apart) {
  // BGZF slabs are cut by member count, 96 per MB: 80 MB = 7680 members = 30 of the 32 a CU holds of the inflate kernel (64 VGPRs,
  // 78 SGPRs, 4928 B of LDS); 64 MB = 6144 = 24 per CU, which leaves a quarter of the wave slots to the parse kernels of the
  // previous slab.  Inputs whose records are split and parsed by kernels of their own (VCF, BAM, BCF, SAM) take 64 since the end of
  // round 5 -- with the inflate twice as fast as in round 4 those kernels are a quarter of a slab's work, and waiting for slots they
  // set the period: .vcf.gz 36.4 -> 34.5 ms, BAM 39.2 -> 36.7, same box, three passes each -- FASTQ, whose histogram reads the text in
  // place and whose scans are as much reader- as kernel-bound, keeps 80 (90.4 against 91.3-100.9 ms): profiles/r5_slab_size.log.
  size_t slab = (size_t)(bgzf_parsed_apart ? 64 : 80) << 20;
  if (const char* v = getenv("EXON_HIP_GPU_PARSE_SLAB_MB")) {
    const long mb = atol(v);
    if (mb >= 1 && mb <= 1024) slab = (size_t)mb << 20;
  }
  return slab;
}

static double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// ---- text slabs in HBM ----------------------------------------------------------------------------------------------
// Feeds the GPU-side parsers with slabs of text resident in HBM.  The consumer reports how many bytes of a slab form
// whole records (the device knows, the host never looks at the text); the tail is carried in front of the next slab.
//   plain text : file -> pinned slab (8-thread pread, filled by a background thread behind a reserved gap) -> HBM on a
//                copy stream of its own, under the parse of the previous slab; the carried tail moves device-to-device
//   BGZF       : compressed file -> pinned -> HBM as it is -> inflated ON THE GPU (inflate.hip) -> text; the carried tail
//                moves device-to-device.  3-5x fewer bytes cross PCIe and no host core inflates anything.
// next() returns 1 for "give up, decode on the host" (a corrupt block, a record larger than the gap).
// The slab buffers are large (pinned host + HBM); allocating and freeing them costs tens of milliseconds, so one set per
// ctx is kept between scans (released by exon_hip_ctx_destroy).
struct SlabBuffers {
  bool bgzf = false;
  size_t hcap = 0, tcap = 0, ccap = 0;
  int max_blocks = 0;
  uint8_t* h_buf[2] = {nullptr, nullptr};
  uint8_t* h_ring = nullptr;  // the pinned staging ring (a few pieces) instead of two slab-sized pinned buffers
  size_t ring_bytes = 0;
  hipEvent_t ev_piece[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint8_t* d_comp[2] = {nullptr, nullptr};
  uint8_t* d_text[2] = {nullptr, nullptr};
  exon_hip_bgzf_block* h_blocks = nullptr;
  exon_hip_bgzf_block* d_blocks = nullptr;
  hipStream_t cs = nullptr, xs = nullptr;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // everything but the streams and events (those fit any scan)
  void free_buffers() {
    if (h_ring) hipHostFree(h_ring);
    h_ring = nullptr;
    ring_bytes = 0;
    for (int k = 0; k < 2; ++k) {
      if (h_buf[k]) hipHostFree(h_buf[k]);
      if (d_comp[k]) hipFree(d_comp[k]);
      if (d_text[k]) hipFree(d_text[k]);
      h_buf[k] = d_comp[k] = d_text[k] = nullptr;
    }
    if (h_blocks) hipHostFree(h_blocks);
    if (d_blocks) hipFree(d_blocks);
    h_blocks = d_blocks = nullptr;
    hcap = tcap = ccap = 0;
    max_blocks = 0;
  }
  void free_all() {
    if (h_ring) hipHostFree(h_ring);
    h_ring = nullptr;
    for (auto& e : ev_piece) {
      if (e) hipEventDestroy(e);
      e = nullptr;
    }
    if (cs) {
      exon_bgzf_forget_stream(cs);
      hipStreamDestroy(cs);
    }
    if (xs) hipStreamDestroy(xs);
    cs = xs = nullptr;
    for (auto& e : ev) {
      if (e) hipEventDestroy(e);
      e = nullptr;
    }
    for (int k = 0; k < 2; ++k) {
      if (h_buf[k]) hipHostFree(h_buf[k]);
      if (d_comp[k]) hipFree(d_comp[k]);
      if (d_text[k]) hipFree(d_text[k]);
      h_buf[k] = d_comp[k] = d_text[k] = nullptr;
    }
    if (h_blocks) hipHostFree(h_blocks);
    if (d_blocks) hipFree(d_blocks);
    h_blocks = d_blocks = nullptr;
  }
};
static std::mutex g_slab_mu;
static std::map<exon_hip_ctx*, SlabBuffers> g_slab_cache;

// Called by exon_hip_ctx_create (EXON_HIP_CTX_PREWARM=0 skips it): the two side streams and the events of the file pipelines
// are made with the context instead of inside its first scan (~17 ms of a first scan's 125: profiles/r4_first_scan.log).
void exon_hip_prewarm_ctx(exon_hip_ctx* ctx) {
  if (const char* v = getenv("EXON_HIP_CTX_PREWARM"))
    if (v[0] == '0') return;
  SlabBuffers b;
  int prio_least = 0, prio_greatest = 0;
  const char* pv = getenv("EXON_HIP_STREAM_PRIORITY");
  const bool use_prio = !(pv && pv[0] == '0') && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess && prio_least != prio_greatest;
  bool ok = (use_prio ? hipStreamCreateWithPriority(&b.cs, hipStreamNonBlocking, prio_least) : hipStreamCreateWithFlags(&b.cs, hipStreamNonBlocking)) == hipSuccess &&
            hipStreamCreateWithFlags(&b.xs, hipStreamNonBlocking) == hipSuccess;
  for (int k = 0; ok && k < 6; ++k) ok = hipEventCreateWithFlags(&b.ev[k], hipEventDisableTiming) == hipSuccess;
  if (!ok) {  // not an error: the first scan makes what is missing
    (void)hipGetLastError();
    b.free_all();
    return;
  }
  std::lock_guard<std::mutex> g(g_slab_mu);
  g_slab_cache[ctx] = b;
}

void exon_hip_release_ctx_caches(exon_hip_ctx* ctx) {
  std::lock_guard<std::mutex> g(g_slab_mu);
  auto it = g_slab_cache.find(ctx);
  if (it == g_slab_cache.end()) return;
  it->second.free_all();
  g_slab_cache.erase(it);
}

class GpuTextSource {
 public:
  // trim_last (BGZF only): inflated bytes to drop behind the last block -- an index chunk ends inside its last block
  // gz: `src` delivers the RAW bytes of a plain-gzip (non-BGZF) file; they cross PCIe as they are and are inflated on the GPU
  // (gzip_stream.hip) straight into the text buffer -- everything downstream is the plain-text mode
  GpuTextSource(exon_hip_ctx* ctx, hipStream_t hs, std::unique_ptr<exon::ByteSource> src, bool bgzf, uint64_t skip_first,
                std::string carry, bool binary = false, bool text_async = false, size_t trim_last = 0, bool gz = false)
      : ctx_(ctx), hs_(hs), src_(std::move(src)), rd_(src_.get()), bgzf_(bgzf), binary_(binary), text_async_(text_async), gz_(gz), skip_(skip_first),
        trim_last_(trim_last), carry_(std::move(carry)) {
    slab_ = gz_ ? gz_slab_bytes() : slab_bytes(bgzf_ && !text_async_);
    ring_geometry();
    local_cpus_ = gpu_local_cpus(ctx_->device);
    rd_.cpus = local_cpus_;
    if (bgzf_) {
      // One wavefront inflates one block and a block takes ~3.5 ms however many run beside it, so a launch wants as many
      // blocks as the chip holds wavefronts of this kernel (24 per CU x 256 CUs) and not one more: slabs are cut by BLOCK
      // COUNT.  The bytes read per slab follow the running average block size.
      target_blocks_ = (int)std::max<size_t>(64, 6144 * (slab_ >> 20) / 64);
      // compressed bytes per slab (+ a carried partial block): room for `target_blocks_` members whatever their ratio.  (Until
      // round 4 the cap was 2 x slab_ = 128 MB: enough for the 6144 members of VCF text (112 MB) or BAM (131 MB), but a
      // .fastq.gz member is ~35 KB, its slabs stopped at 3700 members = 14 waves per CU, and the inflate -- whose throughput
      // grows with the resident waves, profiles/r4_inflate_waves_per_cu.log -- ran at 60 % occupancy.  EXON_HIP_COMP_CAP_MB: A/B.)
      comp_cap_ = (size_t)target_blocks_ * 65536 + (1u << 17);
      if (const char* v = getenv("EXON_HIP_COMP_CAP_MB")) {
        const long mb = atol(v);
        if (mb >= 16 && mb <= 2048) comp_cap_ = ((size_t)mb << 20) + (1u << 17);
      }
      text_cap_ = (size_t)target_blocks_ * 65536;   // inflated bytes per slab
    } else {
      text_cap_ = slab_;
      if (gz_) comp_cap_ = text_cap_;  // (stored blocks: a compressed slab never needs to be larger than the text it may produce)
    }
    gap_ = gz_ ? std::max<size_t>(text_cap_ / 16, 16u << 20) : std::max<size_t>(text_cap_ / 4, 1u << 20);  // room for the carried tail (< 1 record + 1 line)
  }
  // Plain-gzip slabs are cut by OUTPUT bytes: 1 GiB of text per slab (a decode call has fixed costs -- the host's proof of the
  // chunk chain, its round trips --, and its chunk size follows the wavefront slots anyway).  End of round 6
  // (profiles/r6_plain_gzip.log): 20 M-read .fastq.gz 306 ms at 512 MiB, 240 ms at 1 GiB; 100 M-row .vcf.gz 267 / 220 ms; 30 M rows
  // 104 / 97 ms.  EXON_HIP_GZ_SLAB_MB overrides.
  static size_t gz_slab_bytes() {
    size_t slab = (size_t)1024 << 20;
    if (const char* v = getenv("EXON_HIP_GZ_SLAB_MB")) {
      const long mb = atol(v);
      if (mb >= 1 && mb <= 4096) slab = (size_t)mb << 20;
    }
    return slab;
  }
  ~GpuTextSource() {
    const double td0 = now_s();
    if (reader_.joinable()) reader_.join();
    if (gz_pref_.joinable()) gz_pref_.join();
    // fill() starts the next piece's read before the H2D copy and the header walk, either of which may throw: a read may still
    // be in flight into the pinned ring that is handed back to the cache (or freed) below -- wait for it first
    (void)rd_.finish();
    const double td1 = now_s();
    if (xs_) hipStreamSynchronize(xs_);
    if (cs_) hipStreamSynchronize(cs_);
    if (ev_carry_) hipEventDestroy(ev_carry_);
    if (gzs_) {
      if (getenv("EXON_HIP_PIPE_TRACE")) {
        exon_hip_gzip_stats gs;
        if (exon_hip_gzip_stream_get_stats(gzs_, &gs) == EXON_HIP_OK)
          fprintf(stderr, "[exon-hip pipe] gzip on the GPU: %llu calls, %llu chunks, %llu repairs, %llu overflow retries, %llu members, %.1f MB -> %.1f MB; decode calls %.1f ms, reads + H2D %.1f ms\n",
                  (unsigned long long)gs.calls, (unsigned long long)gs.chunks, (unsigned long long)gs.repairs, (unsigned long long)gs.overflow_retries, (unsigned long long)gs.members,
                  gs.comp_bytes / 1e6, gs.out_bytes / 1e6, t_gz_decode_ * 1e3, t_gz_read_ * 1e3);
      }
      exon_hip_gzip_stream_destroy(gzs_);
    }
    if (ev_gz_tail_) hipEventDestroy(ev_gz_tail_);
    if (ev_gz_pref_) hipEventDestroy(ev_gz_pref_);
    if (getenv("EXON_HIP_PIPE_TRACE"))
      fprintf(stderr, "[exon-hip pipe] teardown: reader join %.1f ms, stream sync %.1f ms; reader: busy %.1f ms = file reads %.1f ms + pieces %.1f ms (next piece's start %.1f, copy calls %.1f, header walk %.1f) + "
              "leftovers %.1f ms + the rest; thread start lag %.1f ms; consumer waited %.1f ms for inflates, %.1f ms for the reader\n",
              (td1 - td0) * 1e3, (now_s() - td1) * 1e3, t_fill_ * 1e3, t_read_ * 1e3, t_scan_ * 1e3, t_piece_start_ * 1e3, t_h2d_calls_ * 1e3, t_walk_ * 1e3,
              t_rest_ * 1e3, t_spawn_ * 1e3, t_wait_inflate_ * 1e3, t_wait_reader_ * 1e3);
    SlabBuffers b;
    b.cs = cs_;
    b.xs = xs_;
    for (int k = 0; k < 2; ++k) {
      b.ev[k] = ev_h2d_[k];
      b.ev[2 + k] = ev_done_[k];
      b.ev[4 + k] = ev_free_[k];
    }
    b.bgzf = bgzf_;
    b.hcap = hcap_;
    b.tcap = gap_ + text_cap_;
    b.ccap = comp_cap_;
    b.max_blocks = max_blocks_;
    for (int k = 0; k < 2; ++k) {
      b.h_buf[k] = h_buf_[k];
      b.d_comp[k] = d_comp_[k];
      b.d_text[k] = d_text_[k];
    }
    b.h_blocks = h_blocks_;
    b.d_blocks = d_blocks_;
    b.h_ring = h_ring_;
    b.ring_bytes = (size_t)RING_N * (RING_HEAD + RING_PIECE);
    for (int q = 0; q < RING_MAX; ++q) b.ev_piece[q] = ev_piece_[q];
    if (!complete_) b.free_buffers();  // a failed init: the streams and events still go back
    std::lock_guard<std::mutex> g(g_slab_mu);
    SlabBuffers& slot = g_slab_cache[ctx_];
    slot.free_all();
    slot = b;
  }
  size_t max_text_bytes() const { return gap_ + text_cap_ + 64; }
  double reader_seconds() const { return t_fill_; }

  int init() {
    // BGZF: compressed bytes are staged through a small pinned RING, piece by piece (read -> H2D -> header walk), so the pinned
    // memory a scan needs no longer grows with the slab: 4 x 8 MB instead of 2 x 128 MB, i.e. ~8 ms of page pinning on a
    // fresh context instead of 58 (0.23 ms per MB), and reads, copies and the header walk of one slab overlap piecewise.
    hcap_ = 0;  // (plain text goes through the same ring: no slab-sized pinned buffers in either mode)
    if (bgzf_) max_blocks_ = target_blocks_ + 16;  // a slab never takes more blocks than that (see fill)
    {
      std::lock_guard<std::mutex> g(g_slab_mu);
      auto it = g_slab_cache.find(ctx_);
      if (it != g_slab_cache.end()) {
        SlabBuffers& b = it->second;
        // the streams and events fit any scan (exon_hip_ctx_create makes them up front: the first use of two more hardware
        // queues costs ~17 ms on a fresh context); the buffers only one of the same shape
        for (int q = 0; q < RING_MAX; ++q) {
          ev_piece_[q] = b.ev_piece[q];
          b.ev_piece[q] = nullptr;
        }
        cs_ = b.cs;
        xs_ = b.xs;
        for (int k = 0; k < 2; ++k) {
          ev_h2d_[k] = b.ev[k];
          ev_done_[k] = b.ev[2 + k];
          ev_free_[k] = b.ev[4 + k];
        }
        if (b.bgzf == bgzf_ && b.hcap == hcap_ && b.tcap == gap_ + text_cap_ && b.ccap == comp_cap_ && b.max_blocks == max_blocks_ &&
            b.ring_bytes == (size_t)RING_N * (RING_HEAD + RING_PIECE) && b.d_text[0]) {
          for (int k = 0; k < 2; ++k) {
            h_buf_[k] = b.h_buf[k];
            d_comp_[k] = b.d_comp[k];
            d_text_[k] = b.d_text[k];
          }
          h_blocks_ = b.h_blocks;
          d_blocks_ = b.d_blocks;
          h_ring_ = b.h_ring;
          b.h_ring = nullptr;
          complete_ = true;
        } else {
          b.free_buffers();
        }
        g_slab_cache.erase(it);
      }
    }
    const bool trace = getenv("EXON_HIP_PIPE_TRACE") != nullptr;
    const double ti0 = now_s();
    double t_host = 0, t_dev = 0;
    if (!complete_) {
      // (a fresh context pays ~0.23 ms per MB of pinned memory -- 58 ms for the two 128 MB BGZF staging buffers -- and the
      // runtime serialises allocations: handing the second set to a helper thread moved the wait, it did not shorten it)
      for (int k = 0; k < 2; ++k) {
        const double a0 = now_s();
        if (hcap_ && hipHostMalloc((void**)&h_buf_[k], hcap_) != hipSuccess) return fail(ctx_, EXON_HIP_ENOMEM, "pinned slab buffer of %zu bytes could not be allocated", hcap_);
        if (k == 0 && !h_ring_ && hipHostMalloc((void**)&h_ring_, (size_t)RING_N * (RING_HEAD + RING_PIECE)) != hipSuccess)
          return fail(ctx_, EXON_HIP_ENOMEM, "pinned staging ring could not be allocated");
        const double a1 = now_s();
        if (hipMalloc((void**)&d_text_[k], gap_ + text_cap_ + 256) != hipSuccess || ((bgzf_ || gz_) && hipMalloc((void**)&d_comp_[k], comp_cap_ + (gz_ ? GZ_RESERVE : 0) + 8192) != hipSuccess))
          return fail(ctx_, EXON_HIP_ENOMEM, "slab buffers (%zu bytes of text) could not be allocated", text_cap_);
        t_host += a1 - a0;
        t_dev += now_s() - a1;
      }
      if (bgzf_) {
        if (hipHostMalloc((void**)&h_blocks_, 2 * table_bytes()) != hipSuccess || hipMalloc((void**)&d_blocks_, 2 * table_bytes()) != hipSuccess)
          return fail(ctx_, EXON_HIP_ENOMEM, "BGZF block tables could not be allocated");
      }
      complete_ = true;
    }
    const double ti1 = now_s();
    // the inflate stream runs at the LOWEST priority: its kernels fill every workgroup slot for milliseconds, and the parse
    // kernels of the previous slab (the consumer's stream, highest priority: stream.cpp) are short and on the critical path.
    // (Round 4 tried making the streams -- ~17 ms on a fresh context: the first use of two more hardware queues -- and the first
    // inflate's scratch + code objects on a helper thread while this one pins the ring and reads the first piece: the runtime
    // serialises them with the allocations, the setup stayed at 41-42 ms, and the thread was dropped.)
    int prio_least = 0, prio_greatest = 0;
    const char* pv = getenv("EXON_HIP_STREAM_PRIORITY");
    const bool use_prio = !(pv && pv[0] == '0') && hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) == hipSuccess && prio_least != prio_greatest;
    if (((bgzf_ || gz_) && !cs_ && (use_prio ? hipStreamCreateWithPriority(&cs_, hipStreamNonBlocking, prio_least) : hipStreamCreateWithFlags(&cs_, hipStreamNonBlocking)) != hipSuccess) ||
        (!xs_ && hipStreamCreateWithFlags(&xs_, hipStreamNonBlocking) != hipSuccess))
      return fail(ctx_, EXON_HIP_EDEVICE, "stream creation failed");
    for (int k = 0; k < 2; ++k)
      if (!ev_free_[k] && (hipEventCreateWithFlags(&ev_h2d_[k], hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&ev_done_[k], hipEventDisableTiming) != hipSuccess ||
                           hipEventCreateWithFlags(&ev_free_[k], hipEventDisableTiming) != hipSuccess))
        return fail(ctx_, EXON_HIP_EDEVICE, "event creation failed");
    for (int q = 0; q < RING_N; ++q)
      if (!ev_piece_[q] && hipEventCreateWithFlags(&ev_piece_[q], hipEventDisableTiming) != hipSuccess) return fail(ctx_, EXON_HIP_EDEVICE, "event creation failed");
    if (bgzf_) {
      if (!ev_carry_ && hipEventCreateWithFlags(&ev_carry_, hipEventDisableTiming) != hipSuccess) return fail(ctx_, EXON_HIP_EDEVICE, "event creation failed");
      const double ti2 = now_s();
      fill(0, &f_[0]);
      if (f_[0].err) return rethrow(f_[0].err);
      const double ti3 = now_s();
      int rc = enqueue_inflate(0);
      if (rc) return rc;
      if (trace)
        fprintf(stderr, "[exon-hip pipe] init: pinned allocations %.1f ms (%zu MB), device allocations %.1f ms, tables %.1f ms, streams+events %.1f ms, first fill %.1f ms, first inflate enqueue %.1f ms\n",
                t_host * 1e3, (2 * hcap_ + (size_t)RING_N * (RING_HEAD + RING_PIECE)) >> 20, t_dev * 1e3, (ti1 - ti0 - t_host - t_dev) * 1e3, (ti2 - ti1) * 1e3, (ti3 - ti2) * 1e3, (now_s() - ti3) * 1e3);
      if (!f_[0].eof) reader_ = std::thread([this] { run_on(local_cpus_); fill(1, &f_[1]); });
      return EXON_HIP_OK;
    }
    if (carry_.size() > gap_) return 1;  // the host reader had buffered more than the gap holds: host decoder
    if (gz_) {
      if (hipEventCreateWithFlags(&ev_gz_tail_, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_gz_pref_, hipEventDisableTiming) != hipSuccess)
        return fail(ctx_, EXON_HIP_EDEVICE, "event creation failed");
      // symbol scratch: 2 bytes per byte of a slab's text, three times over -- every chunk has the same share, and a chunk decodes
      // up to the first block boundary behind its range: small chunks overshoot their share by a block
      const int rcg = exon_hip_gzip_stream_create(ctx_, (int64_t)(comp_cap_ + GZ_RESERVE), (int64_t)(6 * text_cap_), &gzs_);
      if (rcg) return rcg;
    }
    fill(0, &cur_);
    k_ = 0;
    return EXON_HIP_OK;
  }

  // Next slab: *d_text is 16-byte aligned.  *n == 0 with *final: end of input.
  int next(const uint8_t** d_text, size_t* n, bool* final) {
    if (bgzf_) return next_bgzf(d_text, n, final);
    if (reader_.joinable()) reader_.join();
    if (started_) {
      cur_ = nxt_;
      nxt_ = Filled();
      k_ ^= 1;
    }
    started_ = true;
    if (cur_.err) {
      try { std::rethrow_exception(cur_.err); } catch (const std::exception& e) {
        fail(ctx_, EXON_HIP_EINVAL, "%s%s", e.what(), gz_ ? " (decoding on the host instead)" : "");
        if (gz_ && getenv("EXON_HIP_PIPE_TRACE")) fprintf(stderr, "[exon-hip pipe] gzip on the GPU gave up: %s\n", e.what());
        return gz_ ? 1 : EXON_HIP_EINVAL;  // gzip on the GPU: whatever does not prove goes to the host reader, which reports what the reference's decoder would
      }
    }
    const int k = k_;
    const bool more = !cur_.eof;
    *final = !more;
    // The fresh bytes are already on their way to d_text_[k] + gap_ (the reader queued that copy on the copy stream the
    // moment it had them, under the parse of the previous slab); the tail carried from the previous slab is copied
    // device-to-device right in front of them.  The first slab carries what the host header reader had buffered.
    size_t front = gap_ - carry_dev_, n_text = carry_dev_ + cur_.n;
    if (carry_dev_) HIP_TRY(ctx_, hipMemcpyAsync(d_text_[k] + front, d_text_[carry_k_] + carry_off_, carry_dev_, hipMemcpyDeviceToDevice, hs_));
    carry_dev_ = 0;
    if (cur_.front_extra) {
      front = gap_ - cur_.front_extra;
      n_text = cur_.front_extra + cur_.n;
    }
    if (gz_ && skip_) {  // first slab: the host reader consumed the header
      if (cur_.n < skip_) return 1;  // the header spans more than one slab: not worth handling here
      front = gap_ + (size_t)skip_;
      n_text = cur_.n - (size_t)skip_;
      skip_ = 0;
    }
    HIP_TRY(ctx_, hipStreamWaitEvent(hs_, ev_h2d_[k], 0));
    if (started_prev_) {  // everything the consumer queued on slab i-1 (and the copy out of it) precedes the reuse of its buffer
      HIP_TRY(ctx_, hipEventRecord(ev_free_[k ^ 1], hs_));
      free_rec_[k ^ 1] = true;
    }
    started_prev_ = true;
    if (more) reader_ = std::thread([this, k] { run_on(local_cpus_); fill(k ^ 1, &nxt_); });  // overlaps with everything the GPU does below
    if (!more && n_text > 0) {  // last line without a terminator (a carried tail never ends in one)
      // the slab's last byte is on the host unless the slab is a carried tail only (cur_.n and front_extra both 0)
      const bool ends_nl = (cur_.n > 0 || cur_.front_extra > 0) && cur_.last_byte == '\n';
      if (!ends_nl) {
        HIP_TRY(ctx_, hipMemsetAsync(d_text_[k] + front + n_text, '\n', 1, hs_));
        ++n_text;
      }
    }
    cur_front_ = front;
    cur_text_ = n_text;
    *d_text = d_text_[k] + front;
    *n = n_text;
    return EXON_HIP_OK;
  }

  // The consumer used `consumed` bytes of the slab returned last; the rest is carried into the next one.
  int release(size_t consumed, bool final) {
    const size_t tail = cur_text_ - consumed;
    if (final) return tail == 0 ? EXON_HIP_OK : 1;  // a partial record at the end of the input: host decoder decides
    if (tail > gap_) return 1;
    // copied device-to-device in front of the next slab's text when that slab is taken
    carry_k_ = bgzf_ ? (int)(idx_ & 1) : k_;
    carry_off_ = cur_front_ + consumed;
    carry_dev_ = tail;
    if (bgzf_) ++idx_;
    return EXON_HIP_OK;
  }

 private:
  size_t table_bytes() const { return (size_t)max_blocks_ * (sizeof(exon_hip_bgzf_block) + sizeof(int)); }
  exon_hip_bgzf_block* h_table(int k) { return reinterpret_cast<exon_hip_bgzf_block*>(reinterpret_cast<uint8_t*>(h_blocks_) + (size_t)k * table_bytes()); }
  exon_hip_bgzf_block* d_table(int k) { return reinterpret_cast<exon_hip_bgzf_block*>(reinterpret_cast<uint8_t*>(d_blocks_) + (size_t)k * table_bytes()); }
  int* h_status(int k) { return reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(h_table(k)) + (size_t)max_blocks_ * sizeof(exon_hip_bgzf_block)); }
  int* d_status(int k) { return reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(d_table(k)) + (size_t)max_blocks_ * sizeof(exon_hip_bgzf_block)); }
  int rethrow(std::exception_ptr e) {
    try { std::rethrow_exception(e); } catch (const std::exception& x) { return fail(ctx_, EXON_HIP_EINVAL, "%s", x.what()); }
    return EXON_HIP_EINVAL;
  }

  // BGZF mode is a two-stream pipeline: while the consumer parses slab i on its stream, slab i+1 crosses PCIe and is
  // inflated on the copy stream `cs_` (the inflated bytes always land at offset gap_ of their text buffer; the tail
  // carried from the previous slab is copied right in front of them when the slab is taken -- the parsers accept any
  // alignment).  ev_done_[k]: inflate of buffer k finished; ev_free_[k]: the consumer is done with buffer k.
  int enqueue_inflate(int k) {
    const Filled& f = f_[k];
    enq_[k] = false;
    if (f.n_blocks == 0) return EXON_HIP_OK;
    if (free_rec_[k]) HIP_TRY(ctx_, hipStreamWaitEvent(cs_, ev_free_[k], 0));
    HIP_TRY(ctx_, hipStreamWaitEvent(cs_, ev_h2d_[k], 0));  // the compressed bytes + the block table (copied by the reader on xs_)
    // the kernels write the per-block status straight into the pinned (device-visible) host words: 4 bytes per block
    int* status_dev = nullptr;
    HIP_TRY(ctx_, hipHostGetDevicePointer((void**)&status_dev, h_status(k), 0));
    HIP_TRY(ctx_, exon_bgzf_inflate_launch(cs_, d_comp_[k], d_table(k), f.n_blocks, d_text_[k], status_dev, true, -1, /*text_like=*/!binary_ && !text_async_));
    HIP_TRY(ctx_, hipEventRecord(ev_done_[k], cs_));
    enq_[k] = true;
    return EXON_HIP_OK;
  }

  int next_bgzf(const uint8_t** d_text, size_t* n, bool* final) {
    const int k = (int)(idx_ & 1);
    const Filled f = f_[k];
    const double tw0 = now_s();
    if (enq_[k]) {
      HIP_TRY(ctx_, hipEventSynchronize(ev_done_[k]));
      t_wait_inflate_ += now_s() - tw0;
      const int* hstat = h_status(k);
      for (int i = 0; i < f.n_blocks; ++i)
        if (hstat[i] != 0) {
          fail(ctx_, EXON_HIP_EINVAL, "BGZF block: %s (decoding on the host instead)", exon_bgzf_status_name(hstat[i]));
          return 1;
        }
    }
    const bool more = !f.eof;
    *final = !more;
    // the tail carried from the previous slab goes right in front of this slab's inflated bytes
    size_t front = gap_ - carry_dev_, n_text = carry_dev_ + f.out_bytes;
    if (text_async_) {
      // the plan's kernel reads the text of slab i-1 in place: its buffer is free when the plan's stream gets here
      if (carry_dev_) HIP_TRY(ctx_, hipMemcpyAsync(d_text_[k] + front, d_text_[carry_k_] + carry_off_, carry_dev_, hipMemcpyDeviceToDevice, hs_));
      if (idx_ > 0) {
        HIP_TRY(ctx_, hipEventRecord(ev_free_[k ^ 1], hs_));
        free_rec_[k ^ 1] = true;
      }
    } else if (carry_dev_) {
      // The parsers return (synchronously) with everything they need copied out of the text, so the buffer of slab i-1
      // is free but for the carried tail.  Copy it on the INFLATE stream: stream order puts it ahead of the inflate that
      // overwrites its source, and it does not queue behind the plan's kernel of slab i-1, which the inflate that is
      // running may starve for milliseconds (the next inflate used to wait for that kernel: a 2.8 ms hole per slab).
      HIP_TRY(ctx_, hipMemcpyAsync(d_text_[k] + front, d_text_[carry_k_] + carry_off_, carry_dev_, hipMemcpyDeviceToDevice, cs_));
      HIP_TRY(ctx_, hipEventRecord(ev_carry_, cs_));
      HIP_TRY(ctx_, hipStreamWaitEvent(hs_, ev_carry_, 0));
    }
    carry_dev_ = 0;
    if (skip_) {  // first slab: the host reader consumed the header
      if (f.out_bytes < skip_) return 1;  // the header spans more than one slab: not worth handling here
      front = gap_ + (size_t)skip_;
      n_text = f.out_bytes - (size_t)skip_;
      skip_ = 0;
    }
    if (!more && trim_last_) {  // an index chunk ends inside its last block: the rest of that block is not part of it
      if (trim_last_ > n_text) return 1;
      n_text -= trim_last_;
    }
    if (more) {  // slab i+1: inflate it while the consumer works on slab i; then let the reader fetch slab i+2
      const double tj0 = now_s();
      if (reader_.joinable()) reader_.join();
      t_wait_reader_ += now_s() - tj0;
      if (f_[k ^ 1].err) return rethrow(f_[k ^ 1].err);
      int rc = enqueue_inflate(k ^ 1);
      if (rc) return rc;
      if (!f_[k ^ 1].eof) {
        t_spawn_at_ = now_s();
        reader_ = std::thread([this, k] { run_on(local_cpus_); fill(k, &f_[k]); });
      }
    }
    if (!more && n_text > 0 && !binary_) {  // last line without a terminator
      uint8_t lastb = 0;
      HIP_TRY(ctx_, hipMemcpyAsync(&lastb, d_text_[k] + front + n_text - 1, 1, hipMemcpyDeviceToHost, hs_));
      HIP_TRY(ctx_, hipStreamSynchronize(hs_));
      if (lastb != '\n') {
        HIP_TRY(ctx_, hipMemsetAsync(d_text_[k] + front + n_text, '\n', 1, hs_));
        ++n_text;
      }
    }
    cur_front_ = front;
    cur_text_ = n_text;
    *d_text = d_text_[k] + front;
    *n = n_text;
    return EXON_HIP_OK;
  }

  struct Filled {
    size_t n = 0;          // plain: fresh text bytes behind the gap; bgzf: compressed bytes of whole blocks
    int n_blocks = 0;      // bgzf
    size_t out_bytes = 0;  // bgzf: inflated size of those blocks
    size_t front_extra = 0;  // plain, first slab: bytes of the host reader's buffer placed in front of the fresh ones
    uint8_t last_byte = 0;   // plain: the slab's last byte (does the input end with a newline?)
    bool eof = false;
    std::exception_ptr err;
  };
  // background thread: next chunk of the file into h_buf_[k]
  void fill(int k, Filled* f) {
    const double t_fill0 = now_s();
    if (t_spawn_at_ > 0) t_spawn_ += t_fill0 - t_spawn_at_;
    struct Acc { double* a; double t0; ~Acc() { *a += now_s() - t0; } } acc{&t_fill_, t_fill0};
    try {
      if (gz_) {
        fill_gz(k, f);
        return;
      }
      if (!bgzf_) {
        // plain text: file -> pinned ring piece -> d_text_[k] behind the gap, piece by piece on the copy stream (under the parse
        // of the previous slab); what the host header reader had buffered goes in front of the first slab
        hipSetDevice(ctx_->device);
        if (free_rec_[k] && hipStreamWaitEvent(xs_, ev_free_[k], 0) != hipSuccess) throw std::runtime_error("H2D of a text slab failed");
        f->front_extra = 0;
        if (!carry_.empty()) {
          if (hipMemcpyAsync(d_text_[k] + gap_ - carry_.size(), carry_.data(), carry_.size(), hipMemcpyHostToDevice, xs_) != hipSuccess ||
              hipStreamSynchronize(xs_) != hipSuccess)  // pageable source: it must stay alive until the copy has been staged
            throw std::runtime_error("H2D of a text slab failed");
          f->front_extra = carry_.size();
          f->last_byte = (uint8_t)carry_.back();
          carry_.clear();
        }
        size_t off = 0;
        bool eof = false;
        struct Piece {
          bool active = false;
          int p = 0;
          uint8_t* base = nullptr;
          size_t want = 0;
        };
        auto start_piece = [&](size_t at_off) {
          Piece pc;
          if (at_off >= text_cap_) return pc;
          pc.p = ring_next_;
          ring_next_ = (pc.p + 1) % RING_N;
          if (piece_used_[pc.p] && hipEventSynchronize(ev_piece_[pc.p]) != hipSuccess) throw std::runtime_error("staging ring: event wait failed");
          pc.base = h_ring_ + (size_t)pc.p * (RING_HEAD + RING_PIECE) + RING_HEAD;
          pc.want = std::min(RING_PIECE, text_cap_ - at_off);
          rd_.begin(pc.base, pc.want);
          pc.active = true;
          return pc;
        };
        Piece cur = start_piece(0);
        while (cur.active) {
          const double tf0 = now_s();
          const size_t got = rd_.finish();
          t_read_ += now_s() - tf0;
          eof = got < cur.want;
          Piece nxt;
          if (!eof) nxt = start_piece(off + got);  // read under this piece's copy
          if (got == 0) break;
          if (hipMemcpyAsync(d_text_[k] + gap_ + off, cur.base, got, hipMemcpyHostToDevice, xs_) != hipSuccess || hipEventRecord(ev_piece_[cur.p], xs_) != hipSuccess)
            throw std::runtime_error("H2D of a text slab failed");
          piece_used_[cur.p] = true;
          f->last_byte = cur.base[got - 1];
          off += got;
          cur = nxt;
        }
        f->n = off;
        f->eof = eof;
        if (hipEventRecord(ev_h2d_[k], xs_) != hipSuccess) throw std::runtime_error("H2D of a text slab failed");
        return;
      }
      // [bytes left over from the previous slab | fresh bytes] -> d_comp_[k], through the pinned ring one piece at a time:
      // the piece's H2D is queued the moment it is read (its own stream, under the inflate of the previous slab) and the
      // header walk of the piece runs under that copy.  Whole blocks only, at most `target` of them, inflated size <= text_cap_.
      // The first slab is an eighth of the others: the GPU starts early, the pipeline is full from the second slab on.
      const int target = first_fill_ ? std::max(64, target_blocks_ / 8) : target_blocks_;
      first_fill_ = false;
      const int cap_blocks = std::min(max_blocks_, target);
      const size_t goal = std::min(comp_cap_, (size_t)((double)est_block_ * target * 1.03) + (1u << 16));
      exon_hip_bgzf_block* hb = h_blocks_tmp(k);
      std::string pend;
      pend.swap(left_);
      size_t pend_pos = 0, off = 0, consumed_total = 0, out_bytes = 0, view_carry = 0;
      int nb = 0;
      bool stop = false;
      std::string rest;  // what follows the last accepted block in the bytes already taken
      hipSetDevice(ctx_->device);
      // piece i+1 is being read (SlabReader's pool) while piece i crosses PCIe and its headers are walked
      struct Piece {
        bool active = false, reading = false;
        int p = 0;
        uint8_t* base = nullptr;
        size_t n_pend = 0, want = 0;
      };
      auto start_piece = [&](size_t at_off) {
        Piece pc;
        if (at_off >= goal) return pc;
        pc.p = ring_next_;
        ring_next_ = (pc.p + 1) % RING_N;
        if (piece_used_[pc.p] && hipEventSynchronize(ev_piece_[pc.p]) != hipSuccess) throw std::runtime_error("staging ring: event wait failed");
        pc.base = h_ring_ + (size_t)pc.p * (RING_HEAD + RING_PIECE) + RING_HEAD;
        pc.want = std::min(RING_PIECE, goal - at_off);
        if (pend_pos < pend.size()) {
          pc.n_pend = std::min(pc.want, pend.size() - pend_pos);
          memcpy(pc.base, pend.data() + pend_pos, pc.n_pend);
          pend_pos += pc.n_pend;
        }
        if (pc.n_pend < pc.want && !file_eof_) {
          rd_.begin(pc.base + pc.n_pend, pc.want - pc.n_pend);
          pc.reading = true;
        }
        pc.active = pc.reading || pc.n_pend > 0;
        return pc;
      };
      Piece cur = start_piece(0);
      while (cur.active) {
        const double tf0 = now_s();
        size_t n = cur.n_pend;
        if (cur.reading) {
          const size_t got = rd_.finish();
          file_eof_ = got < cur.want - cur.n_pend;
          n += got;
        }
        const double tf1 = now_s();
        t_read_ += tf1 - tf0;
        Piece nxt;
        if (n == cur.want) nxt = start_piece(off + n);  // its read runs under everything below
        if (n == 0) break;
        const double tf2 = now_s();
        t_piece_start_ += tf2 - tf1;
        uint8_t* base = cur.base;
        const int p = cur.p;
        if (hipMemcpyAsync(d_comp_[k] + off, base, n, hipMemcpyHostToDevice, xs_) != hipSuccess || hipEventRecord(ev_piece_[p], xs_) != hipSuccess)
          throw std::runtime_error("H2D of a compressed piece failed");
        piece_used_[p] = true;
        const double tf3 = now_s();
        t_h2d_calls_ += tf3 - tf2;
        // the block that straddles the piece boundary: its head (kept in piece_carry_) goes right in front of this piece's bytes
        uint8_t* view = base - view_carry;
        if (view_carry) memcpy(view, piece_carry_.data(), view_carry);
        const size_t vn = view_carry + n;
        int32_t got_nb = 0;
        size_t consumed = 0, ob = 0;
        if (exon_hip_bgzf_scan(view, vn, out_bytes, hb + nb, cap_blocks - nb, &got_nb, &consumed, &ob) != EXON_HIP_OK)
          throw std::runtime_error(exon_hip_last_error(nullptr));
        t_walk_ += now_s() - tf3;
        // text capacity: keep the blocks that fit
        int keep = 0;
        size_t kept_out = 0, kept_consumed = 0;
        while (keep < got_nb && (size_t)hb[nb + keep].out_offset + hb[nb + keep].out_size <= text_cap_) {
          kept_out += hb[nb + keep].out_size;
          ++keep;
        }
        const uint32_t view_global = (uint32_t)(off - view_carry);  // where `view` starts inside d_comp_[k]
        if (keep < got_nb) {  // the slab's text is full: everything from the first rejected block on belongs to the next slab
          kept_consumed = keep ? (size_t)hb[nb + keep - 1].comp_offset + hb[nb + keep - 1].comp_size + 8 : 0;
          stop = true;
        } else {
          kept_consumed = consumed;
          if (nb + keep >= cap_blocks) stop = true;
        }
        for (int i = 0; i < keep; ++i) hb[nb + i].comp_offset += view_global;
        nb += keep;
        out_bytes += kept_out;
        consumed_total = (size_t)view_global + kept_consumed;
        off += n;
        const double tf4 = now_s();
        t_scan_ += tf4 - tf1;
        struct AccRest { double* a; double t0; ~AccRest() { *a += now_s() - t0; } } acc_rest{&t_rest_, tf4};
        const size_t tail = vn - kept_consumed;
        if (stop) {
          rest.assign(reinterpret_cast<const char*>(view) + kept_consumed, tail);
          if (nxt.active) {  // the slab filled up while the next piece was being read: those bytes open the next slab
            size_t m = nxt.n_pend;
            if (nxt.reading) {
              const size_t got = rd_.finish();
              file_eof_ = got < nxt.want - nxt.n_pend;
              m += got;
            }
            rest.append(reinterpret_cast<const char*>(nxt.base), m);
          }
          break;
        }
        if (tail > RING_HEAD) throw std::runtime_error("BGZF block larger than the staging headroom");
        piece_carry_.assign(view + kept_consumed, view + vn);
        view_carry = tail;
        cur = nxt;
      }
      if (!stop && view_carry) rest.assign(reinterpret_cast<const char*>(piece_carry_.data()), view_carry);
      // readable zero padding behind the last byte (the bit reader looks ahead)
      if (off > 0 && hipMemsetAsync(d_comp_[k] + off, 0, 4096, xs_) != hipSuccess) throw std::runtime_error("padding of a compressed slab failed");
      if (nb > 0) est_block_ = (double)consumed_total / nb;
      left_ = rest;
      if (pend_pos < pend.size()) left_.append(pend, pend_pos, std::string::npos);
      if (nb == 0 && off > 0 && !(file_eof_ && left_.empty())) {
        if (file_eof_) throw std::runtime_error("truncated BGZF block at the end of the file");
        throw std::runtime_error("BGZF block larger than the slab");
      }
      const size_t have = off;
      f->n = consumed_total;
      f->n_blocks = nb;
      f->out_bytes = out_bytes;
      f->eof = file_eof_ && left_.empty();
      // The block table follows the compressed bytes on the SAME copy stream and the event behind it covers both: the inflate
      // stream then holds kernels only.  (The table used to cross on the inflate stream, and the status words came back on it
      // behind the kernels: copies that wait for kernels sit at the head of a DMA queue, and the next slab's 112 MB copy,
      // queued behind them by the runtime, did not start until the inflate had finished -- the timeline showed the inflate
      // stream idle for 2 ms per slab.)  The pinned table of buffer k is free: the inflate of the slab two back has finished.
      if (xs_) {
        exon_hip_bgzf_block* pt = h_table(k);
        for (int i = 0; i < nb; ++i) {
          pt[i] = hb[i];
          pt[i].out_offset += (uint32_t)gap_;
        }
        if ((nb > 0 && hipMemcpyAsync(d_table(k), pt, (size_t)nb * sizeof(exon_hip_bgzf_block), hipMemcpyHostToDevice, xs_) != hipSuccess) ||
            (have > 0 && hipEventRecord(ev_h2d_[k], xs_) != hipSuccess))
          throw std::runtime_error("H2D of a block table failed");
      }
    } catch (...) {
      f->err = std::current_exception();
    }
  }
  // Plain gzip: [compressed tail of the previous slab (device to device) | fresh bytes through the pinned ring] -> d_comp_, inflated by
  // exon_hip_gzip_stream_decode on the copy stream straight into d_text_[k] behind the gap.  The decode takes whole DEFLATE blocks
  // while their text fits; what it did not use opens the next slab.  The compressed bytes per slab follow the running ratio.
  // While a slab is being decoded a helper thread already reads the NEXT slab's fresh bytes and copies them (on a stream of their
  // own) behind a reserve at the front of the other compressed buffer; the tail of this slab is then placed right in front of them.
  static constexpr size_t GZ_RESERVE = 32u << 20;
  // file -> pinned ring -> d_comp_[kc] + at, up to `want` bytes, copies on stream `cp`; returns the bytes that arrived
  size_t gz_read(int kc, size_t at, size_t want, hipStream_t cp) {
    struct Piece {
      bool active = false;
      int p = 0;
      uint8_t* base = nullptr;
      size_t want = 0;
    };
    size_t got_total = 0;
    auto start_piece = [&](size_t done) {
      Piece pc;
      if (done >= want || file_eof_) return pc;
      pc.p = ring_next_;
      ring_next_ = (pc.p + 1) % RING_N;
      if (piece_used_[pc.p] && hipEventSynchronize(ev_piece_[pc.p]) != hipSuccess) throw std::runtime_error("staging ring: event wait failed");
      pc.base = h_ring_ + (size_t)pc.p * (RING_HEAD + RING_PIECE) + RING_HEAD;
      pc.want = std::min(RING_PIECE, want - done);
      rd_.begin(pc.base, pc.want);
      pc.active = true;
      return pc;
    };
    Piece cur = start_piece(0);
    size_t begun = cur.active ? cur.want : 0;
    while (cur.active) {
      const size_t got = rd_.finish();
      if (got < cur.want) file_eof_ = true;
      Piece nxt;
      if (!file_eof_) {
        nxt = start_piece(begun);
        if (nxt.active) begun += nxt.want;
      }
      if (got == 0) break;
      if (hipMemcpyAsync(d_comp_[kc] + at + got_total, cur.base, got, hipMemcpyHostToDevice, cp) != hipSuccess || hipEventRecord(ev_piece_[cur.p], cp) != hipSuccess)
        throw std::runtime_error("H2D of a compressed piece failed");
      piece_used_[cur.p] = true;
      got_total += got;
      cur = nxt;
    }
    return got_total;
  }
  void fill_gz(int k, Filled* f) {
    hipSetDevice(ctx_->device);
    if (free_rec_[k] && hipStreamWaitEvent(xs_, ev_free_[k], 0) != hipSuccess) throw std::runtime_error("gzip slab: event wait failed");
    f->front_extra = 0;
    const int kc = (int)(gz_fills_++ & 1);
    // the helper's bytes: d_comp_[kc] + GZ_RESERVE ..
    size_t pre = 0;
    if (gz_pref_.joinable()) {
      gz_pref_.join();
      if (gz_pref_err_) std::rethrow_exception(gz_pref_err_);
      pre = gz_pref_len_;
      gz_pref_len_ = 0;
      if (hipStreamWaitEvent(xs_, ev_gz_pref_, 0) != hipSuccess) throw std::runtime_error("gzip slab: event wait failed");
    }
    size_t start = 0, off = 0;  // the slab's bytes are d_comp_[kc] + [start, off)
    if (pre && gz_tail_len_ <= GZ_RESERVE) {
      start = GZ_RESERVE - gz_tail_len_;
      off = GZ_RESERVE + pre;
      if (gz_tail_len_ && hipMemcpyAsync(d_comp_[kc] + start, d_comp_[kc ^ 1] + gz_tail_off_, gz_tail_len_, hipMemcpyDeviceToDevice, xs_) != hipSuccess)
        throw std::runtime_error("gzip slab: carry copy failed");
    } else {
      if (pre) {  // the tail is larger than the reserve (a slab whose text did not fit): the helper's bytes are read again behind it
        rd_.foff -= (int64_t)pre;
        file_eof_ = false;
      }
      if (gz_tail_len_ && hipMemcpyAsync(d_comp_[kc], d_comp_[kc ^ 1] + gz_tail_off_, gz_tail_len_, hipMemcpyDeviceToDevice, xs_) != hipSuccess)
        throw std::runtime_error("gzip slab: carry copy failed");
      off = gz_tail_len_;
    }
    if (hipEventRecord(ev_gz_tail_, xs_) != hipSuccess) throw std::runtime_error("gzip slab: event record failed");  // (the other buffer may be overwritten behind this)
    size_t goal = std::min(gz_room(), std::max(off - start + (1u << 20), gz_target_));
    bool helper_started = false;
    for (;;) {
      const double tr0 = now_s();
      if (off - start < goal && off < gz_room()) off += gz_read(kc, off, std::min(goal - (off - start), gz_room() - off), xs_);  // (never beyond the buffer)
      if (hipMemsetAsync(d_comp_[kc] + off, 0, 4096, xs_) != hipSuccess) throw std::runtime_error("padding of a compressed slab failed");
      const double tr1 = now_s();
      t_gz_read_ += tr1 - tr0;
      t_read_ += tr1 - tr0;
      if (!file_eof_ && !helper_started && gz_prefetch_enabled()) {
        // the next slab's fresh bytes, under this slab's decode
        helper_started = true;
        const size_t want = std::min(gz_room() - GZ_RESERVE, gz_target_);
        gz_pref_err_ = nullptr;
        gz_pref_ = std::thread([this, kc, want] {
          try {
            run_on(local_cpus_);
            hipSetDevice(ctx_->device);
            if (hipStreamWaitEvent(cs_, ev_gz_tail_, 0) != hipSuccess) throw std::runtime_error("gzip slab: event wait failed");
            gz_pref_len_ = gz_read(kc ^ 1, GZ_RESERVE, want, cs_);
            if (hipEventRecord(ev_gz_pref_, cs_) != hipSuccess) throw std::runtime_error("gzip slab: event record failed");
          } catch (...) {
            gz_pref_err_ = std::current_exception();
          }
        });
      }
      int64_t consumed = 0, produced = 0;
      int32_t ended = 0;
      const bool final_in = !gz_pref_.joinable() && file_eof_;  // (while the helper runs, file_eof_ is its to write)
      const int rc = exon_hip_gzip_stream_decode(gzs_, xs_, d_comp_[kc] + start, (int64_t)(off - start), final_in ? 1 : 0, d_text_[k] + gap_, (int64_t)text_cap_, &consumed, &produced,
                                                 &ended);
      t_gz_decode_ += now_s() - tr1;
      if (rc) throw std::runtime_error(exon_hip_last_error(ctx_));
      if (!ended && consumed == 0 && produced == 0) {
        // not one whole block in these bytes: take more (a block larger than the largest slab is the host reader's)
        if (gz_pref_.joinable()) {  // the helper holds the file's next bytes: they go behind this slab's
          gz_pref_.join();
          if (gz_pref_err_) std::rethrow_exception(gz_pref_err_);
          const size_t m = std::min(gz_pref_len_, gz_room() - off);
          if (m < gz_pref_len_) {
            rd_.foff -= (int64_t)(gz_pref_len_ - m);
            file_eof_ = false;
          }
          if (hipStreamWaitEvent(xs_, ev_gz_pref_, 0) != hipSuccess || (m && hipMemcpyAsync(d_comp_[kc] + off, d_comp_[kc ^ 1] + GZ_RESERVE, m, hipMemcpyDeviceToDevice, xs_) != hipSuccess))
            throw std::runtime_error("gzip slab: carry copy failed");
          off += m;
          gz_pref_len_ = 0;
          goal = std::max(goal, off - start);
          if (m) continue;
        }
        if (file_eof_ || off >= gz_room() || goal >= gz_room() - start) throw std::runtime_error("gzip: a DEFLATE block larger than a slab");
        goal = std::min(gz_room() - start, goal * 2);
        continue;
      }
      gz_tail_off_ = start + (size_t)consumed;
      gz_tail_len_ = off - start - (size_t)consumed;
      if (consumed > 0 && produced > 0) {
        const double ratio = (double)produced / (double)consumed;
        gz_target_ = (size_t)std::min<double>((double)comp_cap_, std::max<double>(4 << 20, 0.9 * (double)text_cap_ / ratio));
      }
      f->n = (size_t)produced;
      f->eof = ended != 0;
      if (f->eof && produced > 0) {
        uint8_t lastb = 0;
        if (hipMemcpyAsync(&lastb, d_text_[k] + gap_ + produced - 1, 1, hipMemcpyDeviceToHost, xs_) != hipSuccess || hipStreamSynchronize(xs_) != hipSuccess)
          throw std::runtime_error("gzip slab: last byte copy failed");
        f->last_byte = lastb;
      }
      if (hipEventRecord(ev_h2d_[k], xs_) != hipSuccess) throw std::runtime_error("gzip slab: event record failed");
      return;
    }
  }
  size_t gz_room() const { return comp_cap_ + GZ_RESERVE; }  // bytes a compressed buffer holds (4096 of padding follow)
  static bool gz_prefetch_enabled() {
    static const bool on = [] {
      const char* v = getenv("EXON_HIP_GZ_PREFETCH");
      return !(v && v[0] == '0');
    }();
    return on;
  }
  exon_hip_gzip_stream* gzs_ = nullptr;
  uint64_t gz_fills_ = 0;
  size_t gz_tail_off_ = 0, gz_tail_len_ = 0;
  size_t gz_target_ = 16u << 20;  // compressed bytes of the next slab (the first one is small: the GPU starts early)
  double t_gz_decode_ = 0, t_gz_read_ = 0;
  std::thread gz_pref_;
  size_t gz_pref_len_ = 0;
  std::exception_ptr gz_pref_err_;
  hipEvent_t ev_gz_tail_ = nullptr, ev_gz_pref_ = nullptr;

  exon_hip_bgzf_block* h_blocks_tmp(int k) {
    scan_tmp_[k].resize((size_t)max_blocks_);
    return scan_tmp_[k].data();
  }

  exon_hip_ctx* ctx_;
  hipStream_t hs_;
  std::unique_ptr<exon::ByteSource> src_;
  SlabReader rd_;
  bool bgzf_, binary_;
  bool text_async_;  // the consumer's kernels read the slab text after the parser has returned (FASTQ views)
  bool gz_ = false;  // plain gzip inflated on the GPU (the text side is the plain mode)
  hipEvent_t ev_carry_ = nullptr;
  uint64_t skip_;
  size_t trim_last_ = 0;
  std::string carry_;      // plain: what the host header reader had buffered (goes in front of the first slab)
  size_t carry_dev_ = 0;   // bytes of the carried tail (lives in d_text_[carry_k_] at carry_off_ until the next slab is taken)
  size_t slab_ = 0, comp_cap_ = 0, text_cap_ = 0, gap_ = 0, hcap_ = 0;
  bool complete_ = false;  // all buffers allocated (only complete sets go back to the cache)
  uint8_t* h_buf_[2] = {nullptr, nullptr};
  static constexpr size_t RING_HEAD = 1u << 17;  // headroom in front of a piece for a block that straddles the piece boundary
  static constexpr int RING_MAX = 8;
  // piece size / count of the pinned staging ring.  Same-box sweep (profiles/r4_ring_sweep.log, warm, against round 3's two
  // slab-sized pinned buffers): BGZF inputs run the same with 8, 16 or 32 MB pieces (65-67 ms for the 100 M-row .vcf.gz, the loop
  // is inflate-bound), plain text gets FASTER with pieces of 16 MB and more (plain VCF 108-113 -> 86-90 ms, plain FASTQ 161 ->
  // 126-128 ms: read, copy and parse now overlap piece by piece instead of slab by slab).  BGZF takes the smallest ring (pinning
  // costs 0.23 ms per MB on a fresh context), plain text 4 x 16 MB.  EXON_HIP_RING_PIECE_MB / EXON_HIP_RING_PIECES override.
  size_t RING_PIECE = 8u << 20;
  int RING_N = 4;
  void ring_geometry() {
    RING_PIECE = (size_t)(bgzf_ ? 8 : 16) << 20;
    RING_N = 4;
    if (const char* v = getenv("EXON_HIP_RING_PIECE_MB")) {
      const long mb = atol(v);
      if (mb >= 1 && mb <= 64) RING_PIECE = (size_t)mb << 20;
    }
    if (const char* v = getenv("EXON_HIP_RING_PIECES")) {
      const int n = atoi(v);
      if (n >= 2 && n <= RING_MAX) RING_N = n;
    }
  }
  uint8_t* h_ring_ = nullptr;
  hipEvent_t ev_piece_[RING_MAX] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool piece_used_[RING_MAX] = {false, false, false, false, false, false, false, false};
  int ring_next_ = 0;
  std::vector<uint8_t> piece_carry_;
  uint8_t* d_comp_[2] = {nullptr, nullptr};
  uint8_t* d_text_[2] = {nullptr, nullptr};
  exon_hip_bgzf_block* h_blocks_ = nullptr;  // pinned: table + status of the slab in flight
  exon_hip_bgzf_block* d_blocks_ = nullptr;
  std::vector<exon_hip_bgzf_block> scan_tmp_[2];
  int max_blocks_ = 0, target_blocks_ = 0;
  double est_block_ = 20000;  // running average of the compressed block size
  bool first_fill_ = true;
  std::string left_;
  bool file_eof_ = false;
  Filled cur_, nxt_;
  Filled f_[2];            // bgzf: what the reader put into host buffer k
  double t_fill_ = 0;      // seconds the reader spent filling host buffers
  double t_read_ = 0, t_scan_ = 0;  // of which: file reads, BGZF header walk (with the copy calls and the next piece's start)
  double t_walk_ = 0, t_h2d_calls_ = 0, t_piece_start_ = 0, t_rest_ = 0, t_spawn_ = 0, t_spawn_at_ = 0;  // finer (EXON_HIP_PIPE_TRACE)
  double t_wait_inflate_ = 0, t_wait_reader_ = 0;  // consumer: blocked on the inflate of the slab it wants / on the reader
  uint64_t idx_ = 0;       // bgzf: index of the slab being consumed
  int carry_k_ = 0;
  size_t carry_off_ = 0;
  hipStream_t cs_ = nullptr, xs_ = nullptr;  // inflate stream, H2D stream
  hipEvent_t ev_h2d_[2] = {nullptr, nullptr}, ev_done_[2] = {nullptr, nullptr}, ev_free_[2] = {nullptr, nullptr};
  bool enq_[2] = {false, false}, free_rec_[2] = {false, false};
  std::thread reader_;
  const cpu_set_t* local_cpus_ = nullptr;
  int k_ = 0;
  bool started_ = false, started_prev_ = false;
  size_t cur_front_ = 0, cur_text_ = 0;
};


// raw bytes [lo, hi) of a file (the BGZF blocks of one index chunk)
class FileRangeSource : public exon::ByteSource {
 public:
  FileRangeSource(const std::string& path, int64_t lo, int64_t hi) : left_((size_t)(hi - lo)) {
    f_ = fopen(path.c_str(), "rb");
    if (!f_ || fseek(f_, (long)lo, SEEK_SET) != 0) {
      if (f_) fclose(f_);
      throw std::runtime_error("cannot open " + path);
    }
  }
  ~FileRangeSource() override {
    if (f_) fclose(f_);
  }
  size_t read(uint8_t* dst, size_t n) override {
    const size_t got = fread(dst, 1, std::min(n, left_), f_);
    left_ -= got;
    return got;
  }

 private:
  FILE* f_ = nullptr;
  size_t left_;
};

// One index chunk [start, end) of BGZF virtual positions as a byte range of whole blocks + what to cut off at both ends
// of the inflated bytes (IndexedBGZFFile -> BGZFIndexedOffsets in the reference: indexed_bgzf_file.rs:129-155; the opener
// seeks to the block and skips to the intra-block offset: indexed_file_opener.rs:114-162).
struct ChunkRange {
  int64_t lo = 0, hi = 0;  // compressed bytes
  uint64_t skip = 0;       // inflated bytes in front of the chunk's first record
  size_t trim = 0;         // inflated bytes behind its last record
};
static std::vector<ChunkRange> plan_chunk_ranges(const std::string& path, const std::vector<exon::Chunk>& chunks) {
  std::vector<ChunkRange> out;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  try {
    std::vector<uint8_t> blk;
    for (const exon::Chunk& c : chunks) {
      if (c.end <= c.start) continue;
      ChunkRange r;
      r.lo = (int64_t)(c.start >> 16);
      r.skip = c.start & 0xFFFF;
      r.hi = (int64_t)(c.end >> 16);
      const size_t ue = (size_t)(c.end & 0xFFFF);
      if (ue) {  // the chunk ends inside the block at r.hi: that block is needed, its tail is not
        if (fseek(f, (long)r.hi, SEEK_SET) != 0) throw std::runtime_error("seek failed: " + path);
        exon::BgzfBlockInfo info;
        if (!exon::read_bgzf_block(f, &blk, &info, path)) throw std::runtime_error("index chunk ends beyond the file: " + path);
        if (ue > info.isize) throw std::runtime_error("index chunk ends beyond its block: " + path);
        r.hi += (int64_t)info.total;
        r.trim = info.isize - ue;
      }
      if (r.hi > r.lo) out.push_back(r);
    }
  } catch (...) {
    fclose(f);
    throw;
  }
  fclose(f);
  return out;
}

// the pushed-down region as (dictionary id, [a, b]) for the device row mask; id < 0: no row can match
static void region_target(const exon_hip_scan* scan, int32_t* id, int64_t* a, int64_t* b, bool* range_form) {
  const exon::Region& rg = scan->region.region;
  const std::vector<std::string>* names = scan->vcf   ? &scan->vcf->header.contigs
                                          : scan->bcf ? &scan->bcf->header.contigs
                                          : scan->bam ? &scan->bam->ref_names
                                                      : &scan->sam->ref_names;
  *id = -1;
  for (size_t i = 0; i < names->size(); ++i)
    if ((*names)[i] == rg.name) *id = (int32_t)i;
  *a = rg.start;
  *b = rg.end;
  *range_form = scan->bam != nullptr || scan->sam != nullptr;
}

// VCF / FASTQ file -> text slabs in HBM (GpuTextSource) -> GPU parser -> fused kernel.  Returns 1 when the device could
// not decide something: the caller restores the state and re-decodes the file on the host.
// FILTER dictionary of the device parser, names in id order
static int gpu_filter_names(exon_hip_scan* scan, std::vector<std::string>* names) {
  names->clear();
  int rc = EXON_HIP_OK;
  if (scan->bcf && scan->bcf_parser) {
    int32_t nf = 0;
    rc = exon_hip_bcf_parser_filters(scan->bcf_parser, nullptr, nullptr, 0, &nf);
    std::vector<int32_t> lists((size_t)std::max(nf, 1) * 8), counts((size_t)std::max(nf, 1));
    if (!rc) rc = exon_hip_bcf_parser_filters(scan->bcf_parser, lists.data(), counts.data(), nf, &nf);
    if (rc) return rc;
    const std::vector<std::string>& strs = scan->bcf->strings();
    for (int32_t i = 0; i < nf; ++i) {
      std::string name;
      for (int32_t k = 0; k < counts[(size_t)i]; ++k) {
        if (k) name += ';';
        name += strs[(size_t)lists[(size_t)i * 8 + (size_t)k]];
      }
      names->push_back(name);
    }
  } else if (scan->vcf && scan->parser) {
    int32_t nf = 0;
    std::vector<char> buf(1 << 20);
    rc = exon_hip_vcf_parser_filters(scan->parser, buf.data(), buf.size(), &nf);
    if (rc) return rc;
    size_t o = 0;
    for (int32_t i = 0; i < nf; ++i) {
      names->emplace_back(buf.data() + o);
      o += names->back().size() + 1;
    }
  }
  return rc;
}

// the value dictionaries of the String INFO keys the device decoded (VCF text): names[k] for scan column 4 + k (empty for other kinds)
static int gpu_info_names(exon_hip_scan* scan, std::vector<std::vector<std::string>>* names) {
  names->clear();
  if (!scan->vcf || !scan->parser) return EXON_HIP_OK;
  const std::vector<exon::InfoSpec>& specs = scan->vcf->info_specs;
  names->resize(specs.size());
  int q = 0;
  std::vector<char> buf;
  for (size_t k = 0; k < specs.size(); ++k) {
    if (specs[k].kind == 'S') continue;
    if (specs[k].kind == 's') {
      int32_t nv = 0;
      buf.resize(2u << 20);
      const int rc = exon_hip_vcf_parser_info_values(scan->parser, q, buf.data(), buf.size(), &nv);
      if (rc) return rc;
      size_t o = 0;
      for (int32_t i = 0; i < nv; ++i) {
        (*names)[k].emplace_back(buf.data() + o);
        o += (*names)[k].back().size() + 1;
      }
    }
    ++q;
  }
  return EXON_HIP_OK;
}

// pinned host blocks that carry a slab's columns: a few are kept for reuse (pinning costs ~0.2 ms per MB); a block goes back
// when the last batch that views it has been released -- which may be long after its scan was closed, hence process-wide
namespace {
std::mutex g_export_pool_mu;
std::vector<std::pair<void*, size_t>> g_export_pool;
// Pinned blocks come in size classes (a quarter of headroom, rounded to 8 MiB): the slabs of one scan differ by a few rows, and with
// exact sizes a slab a little larger than the last found no block to reuse while the pool filled up with blocks a little too small --
// every slab then paid a hipHostMalloc (0.23 ms per MB) and a hipHostFree (the native timer's warm passes were SLOWER than its first:
// profiles/r6_scan_next_native.log).
size_t export_block_class(size_t bytes) { return (bytes + bytes / 4 + (8u << 20) - 1) & ~(size_t)((8u << 20) - 1); }
void* export_block_get(size_t* bytes) {
  {
    std::lock_guard<std::mutex> g(g_export_pool_mu);
    size_t best = g_export_pool.size();
    for (size_t i = 0; i < g_export_pool.size(); ++i)
      if (g_export_pool[i].second >= *bytes && (best == g_export_pool.size() || g_export_pool[i].second < g_export_pool[best].second)) best = i;
    if (best < g_export_pool.size()) {
      void* p = g_export_pool[best].first;
      *bytes = g_export_pool[best].second;
      g_export_pool.erase(g_export_pool.begin() + (long)best);
      return p;
    }
  }
  void* p = nullptr;
  *bytes = export_block_class(*bytes);
  if (hipHostMalloc(&p, *bytes) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void export_block_put(void* p, size_t bytes) {
  void* drop = p;
  {
    std::lock_guard<std::mutex> g(g_export_pool_mu);
    if (g_export_pool.size() < 6) {
      g_export_pool.emplace_back(p, bytes);
      return;
    }
    size_t small = 0;  // full: the smallest block goes
    for (size_t i = 1; i < g_export_pool.size(); ++i)
      if (g_export_pool[i].second < g_export_pool[small].second) small = i;
    if (g_export_pool[small].second < bytes) {
      drop = g_export_pool[small].first;
      g_export_pool[small] = {p, bytes};
    }
  }
  hipHostFree(drop);
}
}  // namespace

// One slab's device columns -> one pinned host block -> batch_size-row Arrow batches on the exporter's queue.  Without a region
// mask the batches are VIEWS into the block (children with an offset; validity bitmaps shared, null counts left to the
// consumer); with one the kept rows are gathered.  Returns 2 when the consumer has gone away (scan closed with batches left).
// the projected string / list columns of a slab, copied back into ONE pinned block of the export pool (pageable destinations are
// staged by the runtime at a tenth of the link's rate): per-batch arrays are cut out of these
template <class T>
struct Span {
  const T* p = nullptr;
  size_t n = 0;
  const T& operator[](size_t i) const { return p[i]; }
  const T* data() const { return p; }
  const T* begin() const { return p; }
};
struct HostText {
  bool vcf = false, bam = false, bcf = false;
  uint64_t projection = 0;
  Span<int32_t> alt_item_off;  // BCF: off[1] = alt's list offsets, alt_item_off / val[1] its items
  Span<int32_t> off[3], item_off, qual_off;  // qual_off: quality_scores' own list offsets (SAM), else off[2]
  Span<uint8_t> val[3], valid[2];
  Span<int64_t> qual;
  Span<int32_t> zeros;  // n_rows + 1 zero offsets: the item-less `alt` lists
  void* blk = nullptr;
  size_t blk_bytes = 0;
  exon::SharedBlock* sb = nullptr;  // the block, shared with the batches that are views into it
  ~HostText() { exon::block_unref(sb); }
};
// EXON_HIP_PIPE_TRACE: where a slab's export spends its time (seconds; per producer thread: every scan's pipeline runs in its own)
static thread_local double g_t_batches = 0, g_t_release = 0, g_t_text_batch = 0, g_t_views = 0;
static thread_local double g_t_text_kernels = 0, g_t_fetch_text = 0, g_t_fetch_cols = 0, g_t_block_get = 0, g_t_enqueue = 0, g_t_names = 0;
// (the copies go through `cp`: they have arrived when the slab's copy event has fired)
// zero offsets for the item-less `alt` lists of a batch: every batch of up to K_ZERO_ROWS rows points at the same static array
static constexpr int64_t K_ZERO_ROWS = 65000;
static const int32_t k_zero_offsets[K_ZERO_ROWS + 64] = {0};
static int fetch_text(exon_hip_ctx* ctx, SlabCopier* cp, size_t also_reserve, int64_t n_rows, uint64_t projection, const ExonVcfText* vt, const ExonBamText* bt, HostText* h,
                      bool big_batches, const ExonBcfText* ct = nullptr) {
  h->projection = projection;
  struct Want {
    std::function<void(const uint8_t*)> place;  // points the span at its bytes inside the block
    const void* src;
    size_t count, elem;
  };
  std::vector<Want> wants;
  auto get = [&](auto& span, const void* src, size_t count) {
    typedef typename std::remove_reference<decltype(*span.p)>::type T;
    auto* sp = &span;
    wants.push_back(Want{[sp, count](const uint8_t* at) { sp->p = reinterpret_cast<const T*>(at); sp->n = count; }, src, count, sizeof(T)});
  };
  const size_t n = (size_t)n_rows, nb = (n + 7) / 8;
  if (vt) {
    h->vcf = true;
    if (projection & EXON_HIP_PROJECT_VCF_ID) {
      get(h->off[0], vt->id_list_offsets, n + 1);
      get(h->valid[0], vt->id_valid, nb);
      get(h->item_off, vt->id_item_offsets, (size_t)vt->n_id_items + 1);
      get(h->val[0], vt->id_values, (size_t)vt->n_id_bytes);
    }
    if (projection & EXON_HIP_PROJECT_VCF_REF) {
      get(h->off[2], vt->ref_offsets, n + 1);
      get(h->val[2], vt->ref_values, (size_t)vt->n_ref_bytes);
    }
    if (projection & EXON_HIP_PROJECT_VCF_ALT) {
      get(h->valid[1], vt->alt_valid, nb);
      if (big_batches) get(h->zeros, nullptr, n + 1);  // (no source: cleared below; batches of up to 65 000 rows share a static array)
    }
  }
  if (ct) {  // BCF: lists with their items, never NULL (eager_array_builder.rs:112-134)
    h->bcf = true;
    if (projection & EXON_HIP_PROJECT_VCF_ID) {
      get(h->off[0], ct->id_list_offsets, n + 1);
      get(h->item_off, ct->id_item_offsets, (size_t)ct->n_id_items + 1);
      get(h->val[0], ct->id_values, (size_t)ct->n_id_bytes);
    }
    if (projection & EXON_HIP_PROJECT_VCF_REF) {
      get(h->off[2], ct->ref_offsets, n + 1);
      get(h->val[2], ct->ref_values, (size_t)ct->n_ref_bytes);
    }
    if (projection & EXON_HIP_PROJECT_VCF_ALT) {
      get(h->off[1], ct->alt_list_offsets, n + 1);
      get(h->alt_item_off, ct->alt_item_offsets, (size_t)ct->n_alt_items + 1);
      get(h->val[1], ct->alt_values, (size_t)ct->n_alt_bytes);
    }
  }
  if (bt) {
    h->bam = true;
    if (projection & EXON_HIP_PROJECT_BAM_NAME) {
      get(h->off[0], bt->name_offsets, n + 1);
      get(h->val[0], bt->name_values, (size_t)bt->n_name_bytes);
      get(h->valid[0], bt->name_valid, nb);
    }
    if (projection & EXON_HIP_PROJECT_BAM_CIGAR) {
      get(h->off[1], bt->cigar_offsets, n + 1);
      get(h->val[1], bt->cigar_values, (size_t)bt->n_cigar_bytes);
    }
    if (projection & (EXON_HIP_PROJECT_BAM_SEQUENCE | EXON_HIP_PROJECT_BAM_QUALITY_SCORES)) get(h->off[2], bt->seq_offsets, n + 1);
    if (projection & EXON_HIP_PROJECT_BAM_SEQUENCE) get(h->val[2], bt->seq_values, (size_t)bt->n_seq_bytes);
    if (projection & EXON_HIP_PROJECT_BAM_QUALITY_SCORES) {
      get(h->qual, bt->qual_values, (size_t)bt->n_qual_items);
      if (bt->qual_offsets != bt->seq_offsets) get(h->qual_off, bt->qual_offsets, n + 1);  // SAM: QUAL may be '*' next to a SEQ
    }
  }
  size_t total = 64;
  for (const Want& w : wants) total += (w.count * w.elem + 63) & ~(size_t)63;
  if (!cp->reserve(total + 256 * wants.size() + also_reserve)) return fail(ctx, EXON_HIP_ENOMEM, "no device staging buffer of %zu bytes for a slab's string columns", total);
  h->blk_bytes = total;
  const double tb0 = now_s();
  h->blk = export_block_get(&h->blk_bytes);
  g_t_block_get += now_s() - tb0;
  if (!h->blk) return fail(ctx, EXON_HIP_ENOMEM, "no pinned block of %zu bytes for a slab's string columns", total);
  h->sb = new exon::SharedBlock();
  h->sb->block = h->blk;
  h->sb->bytes = h->blk_bytes;
  h->sb->put = export_block_put;
  size_t at = 0;
  for (const Want& w : wants) {
    uint8_t* dst = static_cast<uint8_t*>(h->blk) + at;
    w.place(dst);
    if (!w.src) memset(dst, 0, w.count * w.elem);
    if (w.count && w.src) cp->add(dst, w.src, w.count * w.elem);
    at += (w.count * w.elem + 63) & ~(size_t)63;
  }
  if (cp->err != hipSuccess) return fail(ctx, EXON_HIP_EDEVICE, "string columns of a slab towards the host: %s", hipGetErrorString(cp->err));
  return EXON_HIP_OK;
}
// the projected columns of the rows `rows[0 .. n)` of the slab (in the order of the projection bits), appended to `kids`
// Without a row list (no pushed-down region) the batch's columns are VIEWS into the slab's pinned block: the slab-wide validity /
// offsets / data buffers with ArrowArray::offset = the batch's first row, the item arrays of the lists shared by reference -- no
// per-row work on the host (profiles/r6_scan_next_native.log).
static void text_batch(const HostText& h, const int64_t* rows, int64_t r0, int64_t n, std::vector<struct ArrowArray*>* kids, exon::BatchArena* arena = nullptr) {
  if (!rows) {
    auto utf8_view = [&](const Span<int32_t>& off, const Span<uint8_t>& val, const Span<uint8_t>* valid, int64_t first, int64_t len) {
      return exon::arena_array(arena, len, first, valid ? -1 : 0, 3, valid ? (const void*)valid->data() : nullptr, off.data(), val.data());
    };
    if (h.vcf) {
      if (h.projection & EXON_HIP_PROJECT_VCF_ID) {
        struct ArrowArray* items = utf8_view(h.item_off, h.val[0], nullptr, 0, (int64_t)h.item_off.n - 1);
        kids->push_back(exon::arena_array(arena, n, r0, -1, 2, h.valid[0].data(), h.off[0].data(), nullptr, items));
      }
      if (h.projection & EXON_HIP_PROJECT_VCF_REF) kids->push_back(utf8_view(h.off[2], h.val[2], nullptr, r0, n));
      if (h.projection & EXON_HIP_PROJECT_VCF_ALT) {
        if (h.zeros.p) {  // (batches larger than the static zero array: slab-wide zeros)
          struct ArrowArray* items = exon::arena_array(arena, 0, 0, 0, 3, nullptr, h.zeros.data(), h.zeros.data());
          kids->push_back(exon::arena_array(arena, n, r0, -1, 2, h.valid[1].data(), h.zeros.data(), nullptr, items));
        } else {  // the bitmap from the byte the batch starts in, the offsets from the shared zeros
          struct ArrowArray* items = exon::arena_array(arena, 0, 0, 0, 3, nullptr, k_zero_offsets, k_zero_offsets);
          kids->push_back(exon::arena_array(arena, n, r0 & 7, -1, 2, h.valid[1].data() + (r0 >> 3), k_zero_offsets, nullptr, items));
        }
      }
    }
    if (h.bcf) {
      if (h.projection & EXON_HIP_PROJECT_VCF_ID) {
        struct ArrowArray* items = utf8_view(h.item_off, h.val[0], nullptr, 0, (int64_t)h.item_off.n - 1);
        kids->push_back(exon::arena_array(arena, n, r0, 0, 2, nullptr, h.off[0].data(), nullptr, items));
      }
      if (h.projection & EXON_HIP_PROJECT_VCF_REF) kids->push_back(utf8_view(h.off[2], h.val[2], nullptr, r0, n));
      if (h.projection & EXON_HIP_PROJECT_VCF_ALT) {
        struct ArrowArray* items = utf8_view(h.alt_item_off, h.val[1], nullptr, 0, (int64_t)h.alt_item_off.n - 1);
        kids->push_back(exon::arena_array(arena, n, r0, 0, 2, nullptr, h.off[1].data(), nullptr, items));
      }
    }
    if (h.bam) {
      if (h.projection & EXON_HIP_PROJECT_BAM_NAME) kids->push_back(utf8_view(h.off[0], h.val[0], &h.valid[0], r0, n));
      if (h.projection & EXON_HIP_PROJECT_BAM_CIGAR) kids->push_back(utf8_view(h.off[1], h.val[1], nullptr, r0, n));
      if (h.projection & EXON_HIP_PROJECT_BAM_SEQUENCE) kids->push_back(utf8_view(h.off[2], h.val[2], nullptr, r0, n));
      if (h.projection & EXON_HIP_PROJECT_BAM_QUALITY_SCORES) {
        struct ArrowArray* items = exon::arena_array(arena, (int64_t)h.qual.n, 0, 0, 2, nullptr, h.qual.data(), nullptr);
        kids->push_back(exon::arena_array(arena, n, r0, 0, 2, nullptr, (h.qual_off.p ? h.qual_off : h.off[2]).data(), nullptr, items));
      }
    }
    return;
  }
  auto row_at = [&](int64_t i) { return rows ? rows[i] : r0 + i; };
  auto bit = [&](const Span<uint8_t>& bm, int64_t r) { return (uint8_t)((bm[(size_t)(r >> 3)] >> (r & 7)) & 1); };
  auto utf8 = [&](const Span<int32_t>& off, const Span<uint8_t>& val, const Span<uint8_t>* valid) {
    exon::Utf8Builder b;
    for (int64_t i = 0; i < n; ++i) {
      const int64_t r = row_at(i);
      if (valid && !bit(*valid, r)) b.append_null();
      else b.append_value(reinterpret_cast<const char*>(val.data()) + off[(size_t)r], (size_t)(off[(size_t)r + 1] - off[(size_t)r]));
    }
    if (!valid) b.valid.clear();
    return b.finish();
  };
  if (h.vcf) {
    if (h.projection & EXON_HIP_PROJECT_VCF_ID) {
      exon::ListUtf8Builder b;
      for (int64_t i = 0; i < n; ++i) {
        const int64_t r = row_at(i);
        if (!bit(h.valid[0], r)) {
          b.append_null();
          continue;
        }
        for (int32_t k = h.off[0][(size_t)r]; k < h.off[0][(size_t)r + 1]; ++k)
          b.items.append_value(reinterpret_cast<const char*>(h.val[0].data()) + h.item_off[(size_t)k], (size_t)(h.item_off[(size_t)k + 1] - h.item_off[(size_t)k]));
        b.close_row();
      }
      b.items.valid.clear();
      kids->push_back(b.finish());
    }
    if (h.projection & EXON_HIP_PROJECT_VCF_REF) kids->push_back(utf8(h.off[2], h.val[2], nullptr));
    if (h.projection & EXON_HIP_PROJECT_VCF_ALT) {
      exon::ListUtf8Builder b;
      for (int64_t i = 0; i < n; ++i) {
        if (bit(h.valid[1], row_at(i))) b.close_row();
        else b.append_null();
      }
      b.items.valid.clear();
      kids->push_back(b.finish());
    }
  }
  if (h.bcf) {
    auto list_of = [&](const Span<int32_t>& list_off, const Span<int32_t>& item_off, const Span<uint8_t>& val) {
      exon::ListUtf8Builder b;
      for (int64_t i = 0; i < n; ++i) {
        const int64_t r = row_at(i);
        for (int32_t k = list_off[(size_t)r]; k < list_off[(size_t)r + 1]; ++k)
          b.items.append_value(reinterpret_cast<const char*>(val.data()) + item_off[(size_t)k], (size_t)(item_off[(size_t)k + 1] - item_off[(size_t)k]));
        b.close_row();
      }
      b.items.valid.clear();
      b.valid.clear();
      return b.finish();
    };
    if (h.projection & EXON_HIP_PROJECT_VCF_ID) kids->push_back(list_of(h.off[0], h.item_off, h.val[0]));
    if (h.projection & EXON_HIP_PROJECT_VCF_REF) kids->push_back(utf8(h.off[2], h.val[2], nullptr));
    if (h.projection & EXON_HIP_PROJECT_VCF_ALT) kids->push_back(list_of(h.off[1], h.alt_item_off, h.val[1]));
  }
  if (h.bam) {
    if (h.projection & EXON_HIP_PROJECT_BAM_NAME) kids->push_back(utf8(h.off[0], h.val[0], &h.valid[0]));
    if (h.projection & EXON_HIP_PROJECT_BAM_CIGAR) kids->push_back(utf8(h.off[1], h.val[1], nullptr));
    if (h.projection & EXON_HIP_PROJECT_BAM_SEQUENCE) kids->push_back(utf8(h.off[2], h.val[2], nullptr));
    if (h.projection & EXON_HIP_PROJECT_BAM_QUALITY_SCORES) {
      exon::ListBuilder<int64_t> b;
      for (int64_t i = 0; i < n; ++i) {
        const int64_t r = row_at(i);
        const Span<int32_t>& qo = h.qual_off.p ? h.qual_off : h.off[2];
        const int32_t a = qo[(size_t)r], z = qo[(size_t)r + 1];
        b.items.values.insert(b.items.values.end(), h.qual.begin() + a, h.qual.begin() + z);
        b.close_row();
      }
      kids->push_back(b.finish());
    }
  }
}

static int export_slab(exon_hip_scan* scan, const exon_hip_column* sc, int64_t n_rows, const uint8_t* row_mask, hipStream_t hs, const ExonVcfText* vt = nullptr,
                       const ExonBamText* bt = nullptr, const std::function<int()>* build_text = nullptr, const ExonBcfText* ct = nullptr) {
  GpuExporter* ex = scan->exporter;
  exon_hip_ctx* ctx = ex->ctx;
  // A pushed-down region: the row mask comes back first.  A slab that keeps nothing sends nothing else; when the kept rows are
  // ONE run of consecutive rows (sorted files: every indexed file, and the reference's own benchmark query,
  // exon-benchmarks/src/main.rs:143-157) only that run's part of every column crosses PCIe and goes out as views like an
  // unfiltered slab; rows kept here and there are gathered from the whole slab.
  std::vector<uint8_t> hmask;
  int64_t run_lo = 0, run_hi = n_rows;  // the span of rows that go out as views ...
  std::vector<std::pair<int64_t, int64_t>> runs;  // ... and the runs inside it (one run = the whole span without a mask)
  bool as_views = !row_mask;
  if (row_mask) {
    hmask.resize((size_t)(n_rows + 7) / 8);
    if (hipMemcpyAsync(hmask.data(), row_mask, hmask.size(), hipMemcpyDeviceToHost, hs) != hipSuccess || hipStreamSynchronize(hs) != hipSuccess)
      return fail(ctx, EXON_HIP_EDEVICE, "row mask of a slab back to the host");
    if (n_rows & 7) hmask.back() &= (uint8_t)((1u << (n_rows & 7)) - 1u);
    // the runs of consecutive kept rows: one for a point region over a sorted file, a handful when reads that reach into the
    // region from the left are interleaved with reads that do not (SemiLazyRecord::intersects is an overlap test)
    int64_t kept = 0;
    bool too_many = false;
    constexpr size_t MAX_RUNS = 256;
    int64_t open_lo = -1;
    for (size_t byte = 0; byte < hmask.size() && !too_many; ++byte) {
      const uint8_t m = hmask[byte];
      if (m == 0xFF) {
        if (open_lo < 0) open_lo = (int64_t)byte * 8;
        kept += 8;
        continue;
      }
      if (m == 0 && open_lo < 0) continue;
      for (int b = 0; b < 8; ++b) {
        const int64_t r = (int64_t)byte * 8 + b;
        if ((m >> b) & 1) {
          if (open_lo < 0) open_lo = r;
          ++kept;
        } else if (open_lo >= 0) {
          runs.emplace_back(open_lo, r);
          open_lo = -1;
          if (runs.size() > MAX_RUNS) too_many = true;
        }
      }
    }
    if (open_lo >= 0) runs.emplace_back(open_lo, n_rows);
    if (too_many) {  // (kept is not complete then: the gather below counts for itself)
      runs.clear();
    } else if (kept == 0) {
      return EXON_HIP_OK;
    }
    static const bool gather_forced = [] {
      const char* v = getenv("EXON_HIP_EXPORT_GATHER");  // A/B: 1 = every filtered slab through the row-by-row gather
      return v && v[0] == '1';
    }();
    if (!too_many && !gather_forced) {
      as_views = true;
      run_lo = runs.front().first;
      run_hi = runs.back().second;
    } else {
      runs.clear();
    }
  }
  // the batches of the slab before this one: its copy ran under this slab's inflate and parse
  if (ex->pending) {
    std::function<int()> emit;
    emit.swap(ex->pending);
    const int rc = emit();
    if (rc) return rc;
  }
  SlabCopier cp(ex, hs);
  ++ex->n_exports;
  auto text_p = std::make_shared<HostText>();
  HostText& text = *text_p;
  const double tc0 = now_s();
  const bool vcf_like = scan->vcf || scan->bcf;
  const std::vector<exon::InfoSpec>* specs = scan->vcf ? &scan->vcf->info_specs : scan->bcf ? &scan->bcf->info_specs : nullptr;
  const int n_cols = vcf_like ? 4 + (int)specs->size() : 5;
  // element widths in the scan's column order (0 = no values: a Flag, whose bitmap is its value)
  std::vector<int> elem((size_t)n_cols, 4);
  if (vcf_like) {
    elem[1] = 8;
    for (size_t k = 0; k < specs->size(); ++k) elem[4 + k] = (*specs)[k].kind == 'b' ? 0 : 4;
  } else {
    elem[1] = 1;
    elem[3] = elem[4] = 8;
  }
  // rows [c_lo, c_hi) of every column come back (c_lo a multiple of 8: bitmaps are cut at a byte)
  const int64_t c_lo = as_views ? (run_lo & ~int64_t(7)) : 0, c_hi = as_views ? run_hi : n_rows, c_n = c_hi - c_lo;
  const size_t nb = ((size_t)(c_n + 7) / 8 + 63) & ~size_t(63);
  std::vector<size_t> voff((size_t)n_cols, 0), boff((size_t)n_cols, 0);
  size_t bytes = 0;
  for (int c = 0; c < n_cols; ++c) {
    voff[(size_t)c] = bytes;
    bytes += (((size_t)c_n * (size_t)elem[(size_t)c]) + 63) & ~size_t(63);
    boff[(size_t)c] = bytes;
    bytes += nb;
  }
  const size_t moff = bytes;
  bytes += nb;
  const size_t path_stage = bytes + 512 * (size_t)n_cols;
  if ((vt || bt || ct) && scan->opt.projection) {
    if (build_text) {  // the device builds *vt / *bt / *ct now
      const int rc = (*build_text)();
      if (rc) return rc;
    }
    const double tf0 = now_s();
    const int rc = fetch_text(ctx, &cp, path_stage, n_rows, scan->opt.projection, vt, bt, &text, scan->opt.batch_size > K_ZERO_ROWS, ct);
    g_t_fetch_text += now_s() - tf0;
    if (rc) return rc;
  } else if (!cp.reserve(path_stage)) {
    return fail(ctx, EXON_HIP_ENOMEM, "no device staging buffer of %zu bytes for a slab's columns", path_stage);
  }
  size_t blk_bytes = bytes;
  uint8_t* blk = static_cast<uint8_t*>(export_block_get(&blk_bytes));
  if (!blk) return fail(ctx, EXON_HIP_ENOMEM, "no pinned block of %zu bytes for a slab's columns", bytes);
  exon::SharedBlock* sb = new exon::SharedBlock();
  sb->block = blk;
  sb->bytes = blk_bytes;
  sb->put = export_block_put;
  std::shared_ptr<exon::SharedBlock> sb_ref(sb, [](exon::SharedBlock* b) { exon::block_unref(b); });  // this slab's own reference
  std::vector<bool> has_bits((size_t)n_cols, false);
  for (int c = 0; c < n_cols; ++c) {
    if (elem[(size_t)c] && sc[c].values)
      cp.add(blk + voff[(size_t)c], static_cast<const uint8_t*>(sc[c].values) + (size_t)c_lo * (size_t)elem[(size_t)c], (size_t)c_n * (size_t)elem[(size_t)c]);
    if (sc[c].validity) {
      cp.add(blk + boff[(size_t)c], sc[c].validity + (c_lo >> 3), (size_t)(c_n + 7) / 8);
      has_bits[(size_t)c] = true;
    }
  }
  if (row_mask && !as_views) memcpy(blk + moff, hmask.data(), hmask.size());
  const hipError_t e = cp.launch();
  if (e != hipSuccess) return fail(ctx, EXON_HIP_EDEVICE, "columns of a slab towards the host: %s", hipGetErrorString(e));
  g_t_fetch_cols += now_s() - tc0;
  const int slot = cp.slot;
  const int64_t bs = scan->opt.batch_size > 0 ? scan->opt.batch_size : 8192;
  // ---- everything below runs when the NEXT slab arrives (or the scan ends): by then the copies have landed -----------------------
  ex->pending = [=]() mutable -> int {
  (void)sb_ref;  // (the slab's own reference to its block lives as long as this closure)
  if (hipEventSynchronize(ex->ev_done[slot]) != hipSuccess) return fail(ctx, EXON_HIP_EDEVICE, "columns of a slab back to the host: the copy failed");
  const HostText& text = *text_p;
  const double tn0 = now_s();
  std::vector<std::string> filters;
  if (vcf_like) {
    const int rc = gpu_filter_names(scan, &filters);
    if (rc) return rc;
  }
  std::vector<std::vector<std::string>> info_names;
  if (scan->vcf) {
    const int rc = gpu_info_names(scan, &info_names);
    if (rc) return rc;
  }
  g_t_names += now_s() - tn0;
  // the dictionaries of this slab's batches: built once, shared by every batch's column
  auto dicts_p = std::make_shared<std::vector<std::shared_ptr<const exon::SharedUtf8>>>((size_t)n_cols);
  std::vector<std::shared_ptr<const exon::SharedUtf8>>& dicts = *dicts_p;
  for (int c = 0; c < n_cols; ++c) {
    if (vcf_like && c == 0) dicts[(size_t)c] = std::make_shared<const exon::SharedUtf8>(scan->vcf ? scan->vcf->chrom_dict.names : scan->bcf->chrom_dict.names);
    else if (vcf_like && c == 3) dicts[(size_t)c] = std::make_shared<const exon::SharedUtf8>(filters);
    else if (scan->vcf && c >= 4 && (size_t)(c - 4) < info_names.size() && (*specs)[(size_t)(c - 4)].kind == 's')
      dicts[(size_t)c] = std::make_shared<const exon::SharedUtf8>(info_names[(size_t)(c - 4)]);
    else if (!vcf_like && c == 2) dicts[(size_t)c] = std::make_shared<const exon::SharedUtf8>(scan->bam_dict_view.names);
  }
  auto dict_of_col = [&](int c) -> struct ArrowArray* { return dicts[(size_t)c] ? exon::shared_utf8_array(dicts[(size_t)c]) : nullptr; };
  auto push_batch = [&](struct ArrowArray* out, int64_t n) -> int {
    const double te0 = now_s();
    std::unique_lock<std::mutex> lk(ex->mu);
    ex->cv_put.wait(lk, [&] { return ex->stop || ex->q.size() < ex->cap; });
    g_t_enqueue += now_s() - te0;
    if (ex->stop) {
      lk.unlock();
      out->release(out);
      free(out);
      return 2;
    }
    ex->q.push_back(out);
    ex->emitted += n;
    lk.unlock();
    ex->cv_get.notify_one();
    return EXON_HIP_OK;
  };
  auto enqueue_arena = [&](exon::BatchArena* arena, const std::vector<struct ArrowArray*>& kids, int64_t n) -> int {
    struct ArrowArray* out = static_cast<struct ArrowArray*>(malloc(sizeof *out));
    exon::make_struct_of_arena(out, n, arena, kids);
    return push_batch(out, n);
  };
  auto enqueue = [&](std::vector<struct ArrowArray*> kids, int64_t n) -> int {
    struct ArrowArray* out = static_cast<struct ArrowArray*>(malloc(sizeof *out));
    exon::make_struct(out, n, std::move(kids));
    const double te0 = now_s();
    std::unique_lock<std::mutex> lk(ex->mu);
    ex->cv_put.wait(lk, [&] { return ex->stop || ex->q.size() < ex->cap; });
    g_t_enqueue += now_s() - te0;
    if (ex->stop) {
      lk.unlock();
      out->release(out);
      free(out);
      return 2;
    }
    ex->q.push_back(out);
    ex->emitted += n;
    lk.unlock();
    ex->cv_get.notify_one();
    return EXON_HIP_OK;
  };
  if (as_views) {
    struct Tm {
      double t0 = now_s();
      ~Tm() { g_t_batches += now_s() - t0; }
    } tm;
    if (runs.empty()) runs.emplace_back(run_lo, run_hi);
    for (const auto& run : runs)
    for (int64_t b0 = run.first; b0 < run.second; b0 += bs) {
      const int64_t n = std::min(run.second, b0 + bs) - b0;
      std::vector<struct ArrowArray*> kids;
      const double tv0 = now_s();
      // every array of the batch out of one allocation (exon::BatchArena): per column its array + its dictionary, the text
      // columns' lists + items, the struct itself
      exon::BatchArena* arena = exon::new_batch_arena(2 * n_cols + 8 + 1, n_cols + 4, sb, text.sb, dicts_p);
      for (int c = 0; c < n_cols; ++c) {
        const void* bits = has_bits[(size_t)c] ? blk + boff[(size_t)c] : nullptr;
        const void* vals = elem[(size_t)c] ? (const void*)(blk + voff[(size_t)c]) : bits;  // a Flag: true where present
        struct ArrowArray* dict = dicts[(size_t)c] ? exon::arena_dictionary(arena, *dicts[(size_t)c]) : nullptr;
        kids.push_back(exon::arena_array(arena, n, b0 - c_lo, bits ? -1 : 0, 2, bits, vals, nullptr, nullptr, dict));
      }
      const double tv1 = now_s();
      g_t_views += tv1 - tv0;
      if (text.vcf || text.bam || text.bcf) text_batch(text, nullptr, b0, n, &kids, arena);
      g_t_text_batch += now_s() - tv1;
      const int rc = enqueue_arena(arena, kids, n);
      if (rc) return rc;
    }
    return EXON_HIP_OK;
  }
  // rows kept by the pushed-down region filter, gathered
  std::vector<int64_t> keep;
  const uint8_t* mask = blk + moff;
  for (int64_t r = 0; r < n_rows; ++r)
    if ((mask[(size_t)(r >> 3)] >> (r & 7)) & 1) keep.push_back(r);
  auto bit = [&](int c, int64_t r) { return !has_bits[(size_t)c] ? (uint8_t)1 : (uint8_t)((blk[boff[(size_t)c] + (size_t)(r >> 3)] >> (r & 7)) & 1); };
  for (int64_t b0 = 0; b0 < (int64_t)keep.size(); b0 += bs) {
    const int64_t n = std::min((int64_t)keep.size(), b0 + bs) - b0;
    std::vector<struct ArrowArray*> kids;
    auto prim = [&](int c, auto tag) {
      typedef decltype(tag) T;
      exon::PrimitiveBuilder<T> pb;
      pb.values.resize((size_t)n);
      pb.valid.resize((size_t)n);
      const T* src = reinterpret_cast<const T*>(blk + voff[(size_t)c]);
      for (int64_t i = 0; i < n; ++i) {
        const int64_t r = keep[(size_t)(b0 + i)];
        pb.values[(size_t)i] = src[r];
        pb.valid[(size_t)i] = bit(c, r);
      }
      kids.push_back(pb.finish(dict_of_col(c)));
    };
    for (int c = 0; c < n_cols; ++c) {
      if (elem[(size_t)c] == 8) prim(c, int64_t());
      else if (elem[(size_t)c] == 1) prim(c, uint8_t());
      else if (elem[(size_t)c] == 4 && vcf_like && (c == 2 || (c >= 4 && (*specs)[(size_t)(c - 4)].kind == 'f'))) prim(c, float());
      else if (elem[(size_t)c] == 4) prim(c, int32_t());
      else {  // Flag -> Boolean: true where present, NULL elsewhere
        std::vector<uint8_t> v((size_t)n);
        for (int64_t i = 0; i < n; ++i) v[(size_t)i] = bit(c, keep[(size_t)(b0 + i)]);
        struct ArrowArray* a = static_cast<struct ArrowArray*>(malloc(sizeof *a));
        exon::make_boolean(a, v, v);
        kids.push_back(a);
      }
    }
    if (text.vcf || text.bam || text.bcf) text_batch(text, keep.data() + b0, 0, n, &kids);
    const int rc = enqueue(std::move(kids), n);
    if (rc) return rc;
  }
  return EXON_HIP_OK;
  };
  if (getenv("EXON_HIP_EXPORT_SYNC") && getenv("EXON_HIP_EXPORT_SYNC")[0] == '1') {  // A/B: no overlap with the next slab
    std::function<int()> emit;
    emit.swap(ex->pending);
    return emit();
  }
  return EXON_HIP_OK;
}

// FASTQ batches from the GPU pipeline: the slab's four Utf8 columns (text_columns.hip: name, description?, sequence,
// quality_scores -- exon-fastq/src/config.rs:79-88) come back into ONE pinned block and go out as batch_size-row views into it
// (slab-wide offsets / data buffers, ArrowArray::offset = the batch's first read).
static int export_fastq_slab(exon_hip_scan* scan, const ExonFastqText& ft, int64_t n_reads, hipStream_t hs) {
  GpuExporter* ex = scan->exporter;
  exon_hip_ctx* ctx = ex->ctx;
  if (ex->pending) {  // the batches of the slab before this one
    std::function<int()> emit;
    emit.swap(ex->pending);
    const int rc = emit();
    if (rc) return rc;
  }
  if (n_reads == 0) return EXON_HIP_OK;
  const size_t n = (size_t)n_reads;
  auto pad = [](size_t b) { return (b + 63) & ~size_t(63); };
  size_t total = 64;
  std::array<size_t, 4> at_off, at_val;
  for (int k = 0; k < 4; ++k) {
    at_off[(size_t)k] = total;
    total += pad((n + 1) * 4);
    at_val[(size_t)k] = total;
    total += pad((size_t)ft.n_bytes[k] + 8);
  }
  const size_t at_valid = total;
  total += pad((n + 7) / 8 + 8);
  SlabCopier cp(ex, hs);
  ++ex->n_exports;
  if (!cp.reserve(total + 4096)) return fail(ctx, EXON_HIP_ENOMEM, "no device staging buffer of %zu bytes for a slab of reads", total);
  size_t blk_bytes = total;
  uint8_t* blk = static_cast<uint8_t*>(export_block_get(&blk_bytes));
  if (!blk) return fail(ctx, EXON_HIP_ENOMEM, "no pinned block of %zu bytes for a slab of reads", total);
  exon::SharedBlock* sb = new exon::SharedBlock();
  sb->block = blk;
  sb->bytes = blk_bytes;
  sb->put = export_block_put;
  std::shared_ptr<exon::SharedBlock> sb_ref(sb, [](exon::SharedBlock* b) { exon::block_unref(b); });
  for (int k = 0; k < 4; ++k) {
    cp.add(blk + at_off[(size_t)k], ft.offsets[k], (n + 1) * 4);
    cp.add(blk + at_val[(size_t)k], ft.values[k], (size_t)ft.n_bytes[k]);
  }
  cp.add(blk + at_valid, ft.desc_valid, (n + 7) / 8);
  const hipError_t e = cp.launch();
  if (e != hipSuccess) return fail(ctx, EXON_HIP_EDEVICE, "reads of a slab towards the host: %s", hipGetErrorString(e));
  const int slot = cp.slot;
  const int64_t bs = scan->opt.batch_size > 0 ? scan->opt.batch_size : 8192;
  ex->pending = [=]() -> int {
    (void)sb_ref;
    if (hipEventSynchronize(ex->ev_done[slot]) != hipSuccess) return fail(ctx, EXON_HIP_EDEVICE, "reads of a slab back to the host: the copy failed");
    for (int64_t b0 = 0; b0 < n_reads; b0 += bs) {
      const int64_t m = std::min(n_reads, b0 + bs) - b0;
      std::vector<struct ArrowArray*> kids;
      exon::BatchArena* arena = exon::new_batch_arena(5, 4, sb, nullptr, nullptr);
      for (int k = 0; k < 4; ++k)
        kids.push_back(exon::arena_array(arena, m, b0, k == 1 ? -1 : 0, 3, k == 1 ? (const void*)(blk + at_valid) : nullptr, blk + at_off[(size_t)k], blk + at_val[(size_t)k]));
      struct ArrowArray* out = static_cast<struct ArrowArray*>(malloc(sizeof *out));
      exon::make_struct_of_arena(out, m, arena, kids);
      std::unique_lock<std::mutex> lk(ex->mu);
      ex->cv_put.wait(lk, [&] { return ex->stop || ex->q.size() < ex->cap; });
      if (ex->stop) {
        lk.unlock();
        out->release(out);
        free(out);
        return 2;
      }
      ex->q.push_back(out);
      ex->emitted += m;
      lk.unlock();
      ex->cv_get.notify_one();
    }
    return EXON_HIP_OK;
  };
  if (getenv("EXON_HIP_EXPORT_SYNC") && getenv("EXON_HIP_EXPORT_SYNC")[0] == '1') {
    std::function<int()> emit;
    emit.swap(ex->pending);
    return emit();
  }
  return EXON_HIP_OK;
}

// the last slab's batches (its copy has nothing left to hide behind); also in front of a hand-over to the host reader, which
// continues behind the rows EMITTED
static int export_flush(exon_hip_scan* scan) {
  GpuExporter* ex = scan->exporter;
  if (!ex || !ex->pending) return EXON_HIP_OK;
  std::function<int()> emit;
  emit.swap(ex->pending);
  return emit();
}

static int consume_text_gpu(exon_hip_stream* st, exon_hip_scan* scan, int64_t* rows_out) {
  const bool trace = getenv("EXON_HIP_PIPE_TRACE") != nullptr;  // phase timings on stderr
  const double t_begin = now_s();
  double t_next = 0, t_parse = 0;
  exon_hip_ctx* ctx = exon_hip_stream_ctx(st);
  hipStream_t hs = (hipStream_t)exon_hip_stream_hip_stream(st);
  const bool is_vcf = scan->vcf != nullptr, is_bam = scan->bam != nullptr, is_bcf = scan->bcf != nullptr, is_sam = scan->sam != nullptr;
  const bool filtered = scan->region.active && (is_vcf || is_bam || is_bcf || is_sam);
  const bool indexed = filtered && scan->region.use_index && (is_vcf || is_bam);
  const bool bgzf = gpu_inflate_enabled() && scan->opt.compression != EXON_HIP_COMPRESSION_NONE &&
                    exon::BgzfParallelSource::is_bgzf(scan->path) && (!is_vcf || indexed || scan->vcf->data_offset() >= 0);
  if ((is_bam || is_bcf || indexed) && !bgzf) return 1;
  // a gzip file that is not BGZF (the `else` arm of the reference's openers, fastq/file_opener.rs:79-92): inflated on the GPU too
  // (gzip_stream.hip) when the host reader can say where the records start in the uncompressed stream; EXON_HIP_GPU_GZIP=0: host zlib
  const bool gz = !bgzf && !indexed && !is_bam && !is_bcf && gpu_inflate_enabled() && gpu_gzip_enabled() && scan->opt.compression != EXON_HIP_COMPRESSION_NONE &&
                  is_plain_gzip(scan->path) && (!is_vcf || scan->vcf->data_offset() >= 0) && (!is_sam || scan->sam->data_offset() >= 0);
  if (!scan->exporter) {
    scan->gpu_inflated = bgzf || gz;
    scan->gpu_decoded = false;
  }

  // ---- the pushed-down region filter: (id, [a, b]) + a row-mask buffer + the count of rows kept -----------------
  int32_t rg_id = -1;
  int64_t rg_a = 0, rg_b = 0;
  bool rg_range = false;
  if (filtered) {
    region_target(scan, &rg_id, &rg_a, &rg_b, &rg_range);
    if (!scan->d_region_pass) {
      HIP_TRY(ctx, hipMalloc((void**)&scan->d_region_pass, 8));
      scan->region_ctx = ctx;
    }
    HIP_TRY(ctx, hipMemsetAsync(scan->d_region_pass, 0, 8, hs));
  }

  // ---- the byte sources: the whole stream, or one per index chunk ------------------------------------------------
  std::vector<ChunkRange> ranges;
  if (indexed) {
    try {
      ranges = plan_chunk_ranges(scan->path, is_vcf ? scan->vcf->planned_chunks : scan->bam->planned_chunks);
    } catch (const std::exception& e) {
      return fail(ctx, EXON_HIP_EINVAL, "%s", e.what());
    }
  }
  const size_t n_sources = indexed ? ranges.size() : 1;

  int64_t total = 0;
  int rc = EXON_HIP_OK;
  double t_init = t_begin, t_reader = 0;
  for (size_t si = 0; si < n_sources && rc == EXON_HIP_OK; ++si) {
    std::unique_ptr<GpuTextSource> src;
    try {
      if (indexed) {
        const ChunkRange& r = ranges[si];
        std::unique_ptr<exon::ByteSource> raw(new FileRangeSource(scan->path, r.lo, r.hi));
        src.reset(new GpuTextSource(ctx, hs, std::move(raw), true, r.skip, std::string(), is_bam, false, r.trim));
      } else if (bgzf) {
        const uint64_t skip = is_vcf   ? (uint64_t)scan->vcf->data_offset()
                              : is_bam ? (uint64_t)scan->bam->data_offset()
                              : is_bcf ? (uint64_t)scan->bcf->data_offset()
                              : is_sam ? (uint64_t)scan->sam->data_offset()
                                       : 0;
        std::unique_ptr<exon::ByteSource> raw(new exon::ByteReader(scan->path, exon::Compression::None));
        src.reset(new GpuTextSource(ctx, hs, std::move(raw), true, skip, std::string(), is_bam || is_bcf, /*text_async=*/!is_vcf && !is_bam && !is_bcf && !is_sam));
      } else if (gz) {
        const uint64_t skip = is_vcf ? (uint64_t)scan->vcf->data_offset() : is_sam ? (uint64_t)scan->sam->data_offset() : 0;
        std::unique_ptr<exon::ByteSource> raw(new exon::ByteReader(scan->path, exon::Compression::None));
        src.reset(new GpuTextSource(ctx, hs, std::move(raw), false, skip, std::string(), false, false, 0, /*gz=*/true));
      } else {
        std::string carry;
        std::unique_ptr<exon::ByteSource> text = is_vcf   ? scan->vcf->take_stream(&carry)
                                                 : is_sam ? scan->sam->take_stream(&carry)
                                                          : scan->fastq->take_stream(&carry);
        if (!text) return fail(ctx, EXON_HIP_ESTATE, "scan already consumed");
        src.reset(new GpuTextSource(ctx, hs, std::move(text), false, 0, std::move(carry)));
      }
    } catch (const std::exception& e) {
      return fail(ctx, EXON_HIP_EINVAL, "%s", e.what());
    }
    rc = src->init();
    if (rc) break;
    const double t_src = now_s();
    if (is_vcf && !scan->parser) {
      std::vector<const char*> names;
      for (const auto& c : scan->vcf->header.contigs) names.push_back(c.c_str());
      std::string keys;  // "name:kind,..." from the header-typed specs of the host reader; String LISTS are not decoded
      for (const auto& sp : scan->vcf->info_specs)
        if (sp.kind != 'S') keys += (keys.empty() ? "" : ",") + sp.name + ":" + std::string(1, sp.kind);
      rc = exon_hip_vcf_parser_create(ctx, names.data(), (int32_t)names.size(), keys.empty() ? nullptr : keys.c_str(),
                                      (int64_t)src->max_text_bytes(), &scan->parser);
      if (rc) break;
      scan->parser_ctx = ctx;
    }
    if (is_bam && !scan->bam_parser) {
      rc = exon_hip_bam_parser_create(ctx, (int32_t)scan->bam->ref_names.size(), (int64_t)src->max_text_bytes(), &scan->bam_parser);
      if (rc) break;
    }
    if (is_bcf && !scan->bcf_parser) {
      rc = exon_hip_bcf_parser_create(ctx, (int32_t)scan->bcf->header.contigs.size(), (int32_t)scan->bcf->strings().size(),
                                      (int32_t)scan->bcf->header.samples.size(), (int32_t)scan->bcf->info_key(), (int64_t)src->max_text_bytes(),
                                      &scan->bcf_parser);
      if (rc) break;
      if (scan->bcf->info_specs.size() > 0) {
        std::vector<int32_t> keys;
        std::string kinds;
        for (size_t k = 0; k < scan->bcf->info_specs.size(); ++k) {
          const char kind = scan->bcf->info_specs[k].kind;
          if (kind == 's' || kind == 'S') continue;  // String / Character keys are not decoded on the device
          keys.push_back(scan->bcf->info_keys()[k]);
          kinds += kind;
        }
        if (!keys.empty()) rc = exon_hip_bcf_parser_set_info_keys(scan->bcf_parser, keys.data(), kinds.c_str(), (int32_t)keys.size());
        if (rc) break;
      }
    }
    if (is_sam && !scan->sam_parser) {
      std::vector<const char*> names;
      for (const auto& c : scan->sam->ref_names) names.push_back(c.c_str());
      rc = exon_hip_sam_parser_create(ctx, names.data(), (int32_t)names.size(), (int64_t)src->max_text_bytes(), &scan->sam_parser);
      if (rc) break;
    }
    if (!is_vcf && !is_bam && !is_bcf && !is_sam && !scan->fq_parser) {
      rc = exon_hip_fastq_parser_create(ctx, (int64_t)src->max_text_bytes(), &scan->fq_parser);
      if (rc) break;
    }
    if (si == 0) {
      t_init = now_s();
      if (trace) fprintf(stderr, "[exon-hip pipe] source init %.1f ms, parser create %.1f ms\n", (t_src - t_begin) * 1e3, (t_init - t_src) * 1e3);
    }
    for (;;) {
      const uint8_t* d_text = nullptr;
      size_t n = 0;
      bool final = false;
      const double t0 = now_s();
      rc = src->next(&d_text, &n, &final);
      const double t1 = now_s();
      t_next += t1 - t0;
      if (rc) break;
      size_t consumed = 0;
      if (n > 0 && (is_vcf || is_bcf || is_bam || is_sam)) {
        // parsed columns in the scan's column order: VCF / BCF 0 chrom 1 pos 2 qual 3 filter 4.. info fields; BAM / SAM 0 flag 1 mapq 2 ref 3 start 4 end
        exon_hip_column sc[4 + EXON_HIP_MAX_INFO_FIELDS];
        memset(sc, 0, sizeof sc);
        int64_t n_rows = 0;
        const int32_t* id_col = nullptr;
        const uint8_t *id_valid = nullptr, *pos_valid = nullptr;
        const int64_t *c_start = nullptr, *c_end = nullptr;
        if (is_vcf || is_bcf) {
          exon_hip_vcf_columns cols;
          // a fused plan that groups by a String key: NULL is a group of its own (the empty text's id); batches keep NULL
          if (is_vcf) exon_hip_vcf_parser_set_null_key(scan->parser, scan->exporter ? 0 : 1);
          rc = is_vcf ? exon_hip_vcf_parser_parse(scan->parser, hs, d_text, (int64_t)n, &cols)
                      : exon_hip_bcf_parser_parse(scan->bcf_parser, hs, d_text, (int64_t)n, &cols);
          t_parse += now_s() - t1;
          if (!rc && cols.n_undecided > 0) rc = 1;
          if (rc) break;
          consumed = (size_t)cols.consumed_bytes;
          n_rows = cols.n_rows;
          sc[0].values = cols.chrom_id;
          sc[1].values = cols.pos;
          sc[1].validity = cols.pos_valid;
          sc[2].values = cols.qual;
          sc[2].validity = cols.qual_valid;
          sc[3].values = cols.filter_id;
          {  // device key q -> scan column 4 + k: String / Character keys were left out, list keys are no plan operands
            const std::vector<exon::InfoSpec>& specs = is_vcf ? scan->vcf->info_specs : scan->bcf->info_specs;
            int q = 0;
            for (size_t k = 0; k < specs.size() && k < (size_t)EXON_HIP_MAX_INFO_FIELDS; ++k) {
              const char kind = specs[k].kind;
              if (kind == 'S' || (kind == 's' && !is_vcf)) continue;  // (not given to the device parser)
              if (q < cols.n_info && !exon::info_kind_is_list(kind)) {
                sc[4 + k].values = cols.infos[q] ? (const void*)cols.infos[q] : (const void*)cols.infos_valid[q];  // a Flag's values ARE its bitmap
                sc[4 + k].validity = cols.infos_valid[q];
                // a String key's dictionary ids: without a NULL in this slab the column goes out as NOT NULL (a fused plan that
                // groups by it refuses nullable ids: there is no NULL group in its state)
                if (kind == 's' && cols.info_nulls[q] == 0 && !scan->exporter) sc[4 + k].validity = nullptr;
              }
              ++q;
            }
          }
          id_col = cols.chrom_id;
          c_start = c_end = cols.pos;
          pos_valid = cols.pos_valid;
        } else {
          exon_hip_bam_columns cols;
          rc = is_bam ? exon_hip_bam_parser_parse(scan->bam_parser, hs, d_text, (int64_t)n, &cols)
                      : exon_hip_sam_parser_parse(scan->sam_parser, hs, d_text, (int64_t)n, &cols);
          t_parse += now_s() - t1;
          if (trace) fprintf(stderr, "[exon-hip pipe] bam slab %zu bytes: rc %d rows %lld undecided %lld consumed %lld\n", n, rc, (long long)cols.n_rows, (long long)cols.n_undecided, (long long)cols.consumed_bytes);
          if (!rc && cols.n_undecided > 0) rc = 1;
          if (rc) break;
          consumed = (size_t)cols.consumed_bytes;
          n_rows = cols.n_rows;
          sc[0].values = cols.flag;
          sc[1].values = cols.mapq;
          sc[1].validity = cols.mapq_valid;
          sc[2].values = cols.ref_id;
          sc[2].validity = cols.ref_valid;
          sc[3].values = cols.start;
          sc[3].validity = cols.pos_valid;
          sc[4].values = cols.end;
          sc[4].validity = cols.pos_valid;
          id_col = cols.ref_id;
          id_valid = cols.ref_valid;
          c_start = cols.start;
          c_end = cols.end;
          pos_valid = cols.pos_valid;
        }
        if (n_rows > 0) {
          for (auto& c : sc) c.length = n_rows;
          const uint8_t* row_mask = nullptr;
          if (filtered) {
            // the per-record interval hit, on the device: mask = (first operand's validity) AND hit
            const size_t need = (size_t)(n_rows + 7) / 8 + 64;
            if (scan->region_mask_cap < need) {
              HIP_TRY(ctx, hipStreamSynchronize(hs));  // the previous slab's kernel may still read the old buffer
              if (scan->d_region_mask) hipFree(scan->d_region_mask);
              scan->d_region_mask = nullptr;
              scan->region_mask_cap = 0;
              const size_t cap = need + need / 2;
              if (hipMalloc((void**)&scan->d_region_mask, cap) != hipSuccess) return fail(ctx, EXON_HIP_ENOMEM, "row mask of %zu bytes", cap);
              scan->region_mask_cap = cap;
            }
            const int first = exon_hip_stream_plan_first_column(st);
            const uint8_t* in_valid = first >= 0 && first < 4 + EXON_HIP_MAX_INFO_FIELDS ? sc[first].validity : nullptr;
            HIP_TRY(ctx, exon::launch_region_mask(hs, rg_range, id_col, id_valid, c_start, c_end, pos_valid, in_valid, n_rows, rg_id, rg_a, rg_b,
                                                  scan->d_region_mask, scan->d_region_pass));
            row_mask = scan->d_region_mask;
          }
          if (scan->exporter && scan->opt.projection && (is_vcf || is_bam || is_sam || is_bcf)) {
            // the reference's string / list columns of this slab, built on the device from the index the parser has just made
            // (by export_slab, once it knows that the slab keeps rows at all)
            ExonVcfText vt;
            ExonBamText bt;
            ExonBcfText ct;
            const std::function<int()> build_text = [&]() -> int {
              const double tk0 = now_s();
              int r;
              if (is_bcf) {
                int64_t undecided = 0;
                r = exon_text_bcf(ctx, hs, &scan->text_scratch, d_text, (int64_t)n, exon_hip_bcf_parser_row_records(scan->bcf_parser), n_rows, scan->opt.projection, &ct, &undecided);
                if (!r && undecided) r = 1;  // an ID / allele that is not a typed string: the host reader reports what it is
              } else if (is_sam) {
                int64_t undecided = 0;
                r = exon_text_sam(ctx, hs, &scan->text_scratch, d_text, (int64_t)n, exon_hip_sam_parser_newlines(scan->sam_parser), n_rows, scan->opt.projection, &bt, &undecided);
                if (!r && undecided) r = 1;  // a CIGAR / QUAL the device would not print the way the reader does: the host reader takes over
              } else {
                r = is_vcf ? exon_text_vcf(ctx, hs, &scan->text_scratch, d_text, (int64_t)n, exon_hip_vcf_parser_newlines(scan->parser), n_rows, scan->opt.projection, &vt)
                           : exon_text_bam(ctx, hs, &scan->text_scratch, d_text, (int64_t)n, exon_hip_bam_parser_row_records(scan->bam_parser), n_rows, scan->opt.projection, &bt);
              }
              g_t_text_kernels += now_s() - tk0;
              return r;
            };
            rc = export_slab(scan, sc, n_rows, row_mask, hs, is_vcf ? &vt : nullptr, (is_bam || is_sam) ? &bt : nullptr, &build_text, is_bcf ? &ct : nullptr);
          } else
          rc = scan->exporter ? export_slab(scan, sc, n_rows, row_mask, hs)
                              : exon_hip_stream_launch_scan_columns(st, sc, 4 + EXON_HIP_MAX_INFO_FIELDS, n_rows, row_mask);
          // the parser's column buffers (and the row mask) are reused by the next slab; the kernel is stream-ordered before that
          total += n_rows;
        }
      } else if (n > 0) {
        exon_hip_fastq_views v;
        rc = exon_hip_fastq_parser_parse(scan->fq_parser, hs, d_text, (int64_t)n, final ? 1 : 0, &v);
        if (!rc && v.n_undecided > 0) rc = 1;
        if (!rc && !final && v.consumed_bytes == 0) rc = 1;  // not one whole record in a slab
        if (rc) break;
        consumed = (size_t)v.consumed_bytes;
        if (v.n_reads > 0 && scan->exporter) {  // batches: the four Utf8 columns, built on the device
          ExonFastqText ft;
          rc = exon_text_fastq(ctx, hs, &scan->text_scratch, &v, (int64_t)n + 16, &ft);
          if (!rc) rc = export_fastq_slab(scan, ft, v.n_reads, hs);
          total += v.n_reads;
        } else if (v.n_reads > 0) {
          rc = exon_hip_stream_launch_views(st, d_text, v);  // asynchronous: overlaps with preparing the next slab
          total += v.n_reads;
        }
      }
      if (rc) break;
      if (!final && n > 0 && consumed == 0) { rc = 1; break; }  // a record larger than a slab
      const double tr0 = now_s();
      rc = src->release(consumed, final);
      g_t_release += now_s() - tr0;
      if (rc || final) break;
    }
    if (hipStreamSynchronize(hs) != hipSuccess && rc == EXON_HIP_OK) rc = fail(ctx, EXON_HIP_EDEVICE, "stream synchronize failed");
    t_reader += src->reader_seconds();
    src.reset();
  }
  if (scan->exporter) {
    if (rc == EXON_HIP_OK || rc == 1) {  // the last slab's batches (a hand-over continues behind the rows EMITTED)
      const int fr = export_flush(scan);
      if (fr) rc = fr;
    } else if (scan->exporter->pending) {  // an error: nothing more goes out; its copy must not outlive its block
      hipStreamSynchronize(scan->exporter->copy_stream);
      scan->exporter->pending = nullptr;
    }
  }
  const double t_loop = now_s();
  if (trace) {
    if (scan->exporter)
      fprintf(stderr, "[exon-hip pipe] export (cumulative over this thread): text kernels %.1f ms, text columns D2H %.1f (pinned block %.1f), path columns D2H %.1f, names %.1f, batches %.1f (path views %.1f, text views %.1f, waiting for the consumer %.1f), slab release %.1f\n",
              g_t_text_kernels * 1e3, g_t_fetch_text * 1e3, g_t_block_get * 1e3, g_t_fetch_cols * 1e3, g_t_names * 1e3, g_t_batches * 1e3, g_t_views * 1e3, g_t_text_batch * 1e3, g_t_enqueue * 1e3, g_t_release * 1e3);
    fprintf(stderr, "[exon-hip pipe] setup %.1f ms, loop %.1f ms (%zu source(s); slabs: wait+H2D+inflate %.1f, parse %.1f, other %.1f; file reader busy %.1f)\n",
            (t_init - t_begin) * 1e3, (t_loop - t_init) * 1e3, n_sources, t_next * 1e3, t_parse * 1e3, (t_loop - t_init - t_next - t_parse) * 1e3, t_reader * 1e3);
  }
  if (rc == EXON_HIP_OK && filtered) {  // the scan emitted the rows that hit the region
    unsigned long long kept = 0;
    HIP_TRY(ctx, hipMemcpy(&kept, scan->d_region_pass, 8, hipMemcpyDeviceToHost));
    total = (int64_t)kept;
  }
  if (rc == EXON_HIP_OK && (is_bcf || (is_vcf && scan->parser))) {  // FILTER dictionary -> scan (names in id order)
    std::vector<std::string> names;
    rc = gpu_filter_names(scan, &names);
    if (!rc) {
      if (scan->exporter) scan->exporter->final_filters.swap(names);  // (adopted by the consumer's thread at the end of the batches)
      else scan->gpu_filter_dict.names.swap(names);
    }
  }
  if (rc == EXON_HIP_OK && is_vcf && scan->parser) {  // String INFO keys: their dictionaries -> scan
    std::vector<std::vector<std::string>> info_names;
    rc = gpu_info_names(scan, &info_names);
    if (!rc) {
      if (scan->exporter) {
        scan->exporter->final_info_names.swap(info_names);
      } else {
        for (size_t k = 0; k < info_names.size() && k < scan->vcf->info_dicts.size(); ++k)
          if (scan->vcf->info_specs[k].kind == 's') scan->vcf->info_dicts[k].names.swap(info_names[k]);
      }
    }
  }
  if (rc == EXON_HIP_OK) {
    if (scan->exporter) {
      scan->exporter->decoded_on_gpu = true;
      scan->exporter->inflated_on_gpu = bgzf || gz;
    } else {
      scan->rows += total;
      scan->gpu_decoded = true;
    }
    if (rows_out) *rows_out = total;
  }
  return rc;
}

extern "C" {

int exon_hip_scan_decoded_on_gpu(exon_hip_scan* scan, int32_t* decoded, int32_t* inflated) {
  if (!scan || !decoded) return fail(nullptr, EXON_HIP_EINVAL, "NULL argument");
  *decoded = scan->gpu_decoded ? 1 : 0;
  if (inflated) *inflated = (scan->gpu_decoded && scan->gpu_inflated) ? 1 : 0;
  return EXON_HIP_OK;
}

static int consume_scan_impl(exon_hip_stream* st, exon_hip_scan* scan, int64_t* rows);

// column of the scan that holds the contig / reference dictionary a region is named in
static int region_dict_column(const exon_hip_scan* s) {
  return (s->format == EXON_HIP_FORMAT_VCF || s->format == EXON_HIP_FORMAT_BCF) ? 0 : 2;
}

}  // extern "C"

// the host reader from the start (a scan opened for the GPU pipeline has read the header only)
// (built into the exporter, NOT into the scan: the consumer thread may be inside exon_hip_scan_schema / _dictionary_* on the
// scan's own reader at this very moment)
static void open_fallback_reader(exon_hip_scan* scan, GpuExporter* ex) {
  const exon::Compression c = scan->opt.compression == EXON_HIP_COMPRESSION_GZIP   ? exon::Compression::Gzip
                              : scan->opt.compression == EXON_HIP_COMPRESSION_NONE ? exon::Compression::None
                                                                                   : exon::Compression::Auto;
  if (scan->vcf) {
    exon::VCFConfig cfg = scan->vcf->config();
    cfg.defer_decode = false;
    cfg.threads = 0;
    ex->fb_vcf.reset(new exon::VCFBatchReader(scan->path, c, cfg));
  } else if (scan->bam) {
    exon::BAMConfig cfg = scan->bam->config();
    cfg.threads = 0;
    ex->fb_bam.reset(new exon::BAMBatchReader(scan->path, cfg));
  } else if (scan->bcf) {
    exon::VCFConfig cfg = scan->bcf->config();
    cfg.threads = 0;
    ex->fb_bcf.reset(new exon::BCFBatchReader(scan->path, cfg));
  } else if (scan->sam) {
    ex->fb_sam.reset(new exon::SAMBatchReader(scan->path, c, scan->sam->config()));
  } else if (scan->fastq) {
    exon::FASTQConfig cfg = scan->fastq->config();
    cfg.defer_decode = false;
    cfg.threads = 0;
    ex->fb_fastq.reset(new exon::FASTQBatchReader(scan->path, c, cfg));
  }
}

static void gpu_export_producer(exon_hip_scan* scan) {
  GpuExporter* ex = scan->exporter;
  int rc = EXON_HIP_OK;
  std::string err;
  exon_hip_plan_desc d;
  memset(&d, 0, sizeof d);
  d.kind = EXON_HIP_PLAN_REGION_COUNT;  // (a carrier for context and stream: its kernel is never launched, export_slab takes the slabs)
  d.region_start = d.region_end = 1;
  d.columns[0] = 0;
  d.columns[1] = 1;
  rc = exon_hip_plan_create(ex->ctx, &d, &ex->plan);
  if (!rc) rc = exon_hip_stream_open(ex->plan, 0, &ex->st);
  if (!rc && (hipStreamCreateWithFlags(&ex->copy_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ex->ev_ready, hipEventDisableTiming) != hipSuccess ||
              hipEventCreateWithFlags(&ex->ev_done[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ex->ev_done[1], hipEventDisableTiming) != hipSuccess))
    rc = fail(ex->ctx, EXON_HIP_EDEVICE, "copy stream / events of the batch export");
  int64_t rows = 0;
  if (!rc) rc = consume_text_gpu(ex->st, scan, &rows);
  if (ex->pending) {  // (an early error return inside the pipeline: see consume_text_gpu's own clean-up for the other exits)
    if (ex->copy_stream) hipStreamSynchronize(ex->copy_stream);
    ex->pending = nullptr;
  }
  if (rc == 1) {
    // the device could not decide something: the host reader goes over the file again and takes over behind the rows emitted
    try {
      open_fallback_reader(scan, ex);
      int64_t skip = 0;
      {
        std::lock_guard<std::mutex> g(ex->mu);
        skip = ex->emitted;
        ex->decoded_on_gpu = false;
        ex->handed_over = true;
      }
      rc = EXON_HIP_OK;
      for (;;) {
        struct ArrowArray* out = static_cast<struct ArrowArray*>(malloc(sizeof *out));
        memset(out, 0, sizeof *out);
        const bool got = ex->fb_vcf ? ex->fb_vcf->read_batch(out) : ex->fb_bam ? ex->fb_bam->read_batch(out) : ex->fb_bcf ? ex->fb_bcf->read_batch(out) : ex->fb_sam ? ex->fb_sam->read_batch(out) : ex->fb_fastq->read_batch(out);
        if (!got) {
          free(out);
          break;
        }
        if (skip >= out->length) {  // emitted by the GPU pipeline already
          skip -= out->length;
          out->release(out);
          free(out);
          continue;
        }
        if (skip > 0) {  // the batch that straddles the hand-over: its tail, as a slice (children share the offset)
          for (int64_t c = 0; c < out->n_children; ++c) {
            out->children[c]->offset += skip;
            out->children[c]->length -= skip;
            out->children[c]->null_count = -1;
          }
          out->length -= skip;
          skip = 0;
        }
        std::unique_lock<std::mutex> lk(ex->mu);
        ex->cv_put.wait(lk, [&] { return ex->stop || ex->q.size() < ex->cap; });
        if (ex->stop) {
          lk.unlock();
          out->release(out);
          free(out);
          break;
        }
        ex->q.push_back(out);
        ex->emitted += out->length;
        lk.unlock();
        ex->cv_get.notify_one();
      }
    } catch (const std::exception& e) {
      rc = EXON_HIP_EINVAL;
      err = e.what();
    }
  } else if (rc == 2) {
    rc = EXON_HIP_OK;  // the consumer closed the scan
  } else if (rc) {
    err = exon_hip_last_error(ex->ctx);
  }
  {
    std::lock_guard<std::mutex> g(ex->mu);
    ex->rc = rc;
    ex->err = err;
    ex->done = true;
  }
  ex->cv_get.notify_all();
}

static int gpu_next(exon_hip_scan* s, struct ArrowArray* out) {
  GpuExporter* ex = s->exporter;
  if (!ex->started) {
    ex->started = true;
    ex->th = std::thread(gpu_export_producer, s);
  }
  struct ArrowArray* a = nullptr;
  {
    std::unique_lock<std::mutex> lk(ex->mu);
    ex->cv_get.wait(lk, [&] { return ex->done || !ex->q.empty(); });
    if (!ex->q.empty()) {
      a = ex->q.front();
      ex->q.pop_front();
    }
  }
  if (a) {
    ex->cv_put.notify_one();
    *out = *a;  // moved
    free(a);
    s->rows += out->length;
    return EXON_HIP_OK;
  }
  // the producer has finished: its verdict, the FILTER dictionary and the "decoded on the GPU" flags become the scan's
  if (ex->th.joinable()) ex->th.join();
  if (ex->handed_over) {
    // the producer thread is gone: the reader that finished the file becomes the scan's (its dictionaries are the ones the
    // last batches were built with; every batch carries its dictionary values itself, so nothing emitted earlier depends on it)
    if (ex->fb_vcf) {
      s->gpu_filter_dict.names = ex->fb_vcf->filter_dict.names;
      s->vcf = std::move(ex->fb_vcf);
    } else if (ex->fb_bcf) {
      s->gpu_filter_dict.names = ex->fb_bcf->filter_dict.names;
      s->bcf = std::move(ex->fb_bcf);
    } else if (ex->fb_bam) {
      s->bam = std::move(ex->fb_bam);
    } else if (ex->fb_sam) {
      s->sam = std::move(ex->fb_sam);
    } else if (ex->fb_fastq) {
      s->fastq = std::move(ex->fb_fastq);
    }
    ex->handed_over = false;
    ex->final_filters.clear();
    ex->final_info_names.clear();
  }
  if (ex->rc) return fail(ex->ctx, ex->rc, "%s", ex->err.c_str());
  if (!ex->final_filters.empty()) s->gpu_filter_dict.names.swap(ex->final_filters);
  if (s->vcf)
    for (size_t k = 0; k < ex->final_info_names.size() && k < s->vcf->info_dicts.size(); ++k)
      if (s->vcf->info_specs[k].kind == 's') s->vcf->info_dicts[k].names.swap(ex->final_info_names[k]);
  s->gpu_decoded = ex->decoded_on_gpu;
  s->gpu_inflated = ex->inflated_on_gpu;
  return 1;
}

static void gpu_export_shutdown(exon_hip_scan* s) {
  GpuExporter* ex = s->exporter;
  {
    std::lock_guard<std::mutex> g(ex->mu);
    ex->stop = true;
  }
  ex->cv_put.notify_all();
  if (ex->th.joinable()) ex->th.join();
  for (struct ArrowArray* a : ex->q) {
    if (a->release) a->release(a);
    free(a);
  }
  if (ex->copy_stream) hipStreamSynchronize(ex->copy_stream);
  ex->pending = nullptr;
  for (int k = 0; k < 2; ++k) {
    if (ex->d_stage[k]) hipFree(ex->d_stage[k]);
    if (ex->ev_done[k]) hipEventDestroy(ex->ev_done[k]);
  }
  if (ex->ev_ready) hipEventDestroy(ex->ev_ready);
  if (ex->copy_stream) hipStreamDestroy(ex->copy_stream);
  if (ex->st) exon_hip_stream_close(ex->st);
  if (ex->plan) exon_hip_plan_destroy(ex->plan);
  delete ex;
  s->exporter = nullptr;
}

extern "C" {

int exon_hip_stream_consume_scan(exon_hip_stream* st, exon_hip_scan* scan, int64_t* rows) {
  if (!st || !scan) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_stream_consume_scan: NULL argument");
  // a region plan fed by files names its contig; every file numbers its contigs in its own header order
  std::string contig;
  if (exon_hip_stream_region_contig(st, &contig)) {
    exon::Dictionary* d = dict_of(scan, region_dict_column(scan));
    if (!d) return fail(exon_hip_stream_ctx(st), EXON_HIP_EINVAL, "this scan has no contig / reference dictionary to resolve '%s' in", contig.c_str());
    const bool fixed = scan->format == EXON_HIP_FORMAT_BAM || scan->format == EXON_HIP_FORMAT_SAM || scan->format == EXON_HIP_FORMAT_CRAM;
    int32_t id = d->find(contig);
    if (id < 0 && !fixed) id = d->lookup_or_insert(contig.data(), contig.size());  // VCF text may name contigs its header lacks
    const int rc0 = exon_hip_stream_set_region_id(st, id < 0 ? -1 : id);  // -1: no reference of that name, no row matches
    if (rc0) return rc0;
  }
  // group keys travel by VALUE: the scan's dictionary ids are re-keyed into the stream's (stream.cpp)
  bool tracked = false, redirected = false;
  int rc = exon_hip_stream_begin_scan(st, &tracked, &redirected);
  if (rc) return rc;
  rc = consume_scan_impl(st, scan, rows);
  const std::vector<std::string>* keys = nullptr;
  if (tracked) {
    if (exon::Dictionary* d = dict_of(scan, exon_hip_stream_plan_column(st, 2))) keys = &d->names;
  }
  const int rc2 = exon_hip_stream_end_scan(st, keys, tracked, redirected, rc == EXON_HIP_OK);
  return rc ? rc : rc2;
}

}  // extern "C"

static int consume_scan_impl(exon_hip_stream* st, exon_hip_scan* scan, int64_t* rows) {
  int64_t n = 0;
  // K4 over a VCF / BCF scan: the compared column and AVG's argument may be typed INFO fields (scan columns 4 ..), whose
  // type the FILE's header decides: Type=Integer -> Int32 values, compared / averaged as integers (schema_builder.rs:197-205)
  if (exon_hip_stream_plan_kind(st) == EXON_HIP_PLAN_CMP_AVG_BY_GROUP && (scan->vcf || scan->bcf)) {
    const std::vector<exon::InfoSpec>& specs = scan->vcf ? scan->vcf->info_specs : scan->bcf->info_specs;
    auto type_of = [&](int col) {
      return col >= 4 && (size_t)(col - 4) < specs.size() && specs[(size_t)(col - 4)].kind == 'i' ? EXON_HIP_X_INT32 : EXON_HIP_X_FLOAT32;
    };
    exon_hip_stream_set_value_types(st, type_of(exon_hip_stream_plan_column(st, 0)), type_of(exon_hip_stream_plan_column(st, 1)));
  }
  if (!scan->gpu_parse && scan->gpu_candidate && scan->rows == 0 && (scan->vcf || scan->bcf)) {
    const std::vector<exon::InfoSpec>& specs = scan->vcf ? scan->vcf->info_specs : scan->bcf->info_specs;
    bool reads_host_only = false;
    for (int a = 0; a < 4; ++a) {
      const int col = exon_hip_stream_plan_column(st, a);
      if (col >= 4 && (size_t)(col - 4) < specs.size() && !info_kind_decoded_on_gpu(scan, specs[(size_t)(col - 4)].kind)) reads_host_only = true;
    }
    if (!reads_host_only) {
      try {  // the host reader goes back to "header only": the bytes are the device's
        if (scan->vcf) {
          const exon::Compression c = scan->opt.compression == EXON_HIP_COMPRESSION_GZIP   ? exon::Compression::Gzip
                                      : scan->opt.compression == EXON_HIP_COMPRESSION_NONE ? exon::Compression::None
                                                                                           : exon::Compression::Auto;
          exon::VCFConfig cfg = scan->vcf->config();
          cfg.defer_decode = true;
          if (wants_gpu_inflate(&scan->opt, scan->path.c_str())) cfg.threads = 1;
          scan->vcf.reset(new exon::VCFBatchReader(scan->path, c, cfg));
        } else {
          exon::VCFConfig cfg = scan->bcf->config();
          cfg.threads = 1;
          scan->bcf.reset(new exon::BCFBatchReader(scan->path, cfg));
        }
        scan->gpu_parse = true;
        scan->gpu_candidate = false;
      } catch (const std::exception& e) {
        return fail(nullptr, EXON_HIP_EINVAL, "%s", e.what());
      }
    }
  }
  if (scan->gpu_parse && (scan->vcf || scan->fastq || scan->bam || scan->bcf || scan->sam)) {
    // speculative GPU decode; when the device cannot decide something, restore the state and fall back to the host decoder
    exon_hip_ctx* ctx = exon_hip_stream_ctx(st);
    void* snap = nullptr;
    const size_t sb = exon_hip_stream_state_bytes(st);
    if (hipMalloc(&snap, sb ? sb : 16) != hipSuccess) return fail(ctx, EXON_HIP_ENOMEM, "state snapshot allocation failed");
    int64_t rows_before = 0;
    int rc = exon_hip_stream_state_copy(st, snap, false, &rows_before);  // flushes rows staged by earlier pushes first
    if (!rc) rc = consume_text_gpu(st, scan, rows);
    if (rc != 1) {
      hipFree(snap);
      return rc;
    }
    rc = exon_hip_stream_state_copy(st, snap, true, &rows_before);
    hipStreamSynchronize((hipStream_t)exon_hip_stream_hip_stream(st));
    hipFree(snap);
    if (rc) return rc;
    if (const char* strict = getenv("EXON_HIP_GPU_PARSE_STRICT"); strict && strict[0] == '1')
      return fail(ctx, EXON_HIP_ESTATE, "%s: the GPU decoders could not decide every record and EXON_HIP_GPU_PARSE_STRICT=1 forbids the host decoders",
                  scan->path.c_str());
    if (scan->parser) {
      exon_hip_vcf_parser_destroy(scan->parser);
      scan->parser = nullptr;
    }
    if (scan->bcf_parser) {
      exon_hip_bcf_parser_destroy(scan->bcf_parser);
      scan->bcf_parser = nullptr;
    }
    scan->gpu_parse = false;
    try {
      const exon::Compression c = scan->opt.compression == EXON_HIP_COMPRESSION_GZIP   ? exon::Compression::Gzip
                                  : scan->opt.compression == EXON_HIP_COMPRESSION_NONE ? exon::Compression::None
                                                                                       : exon::Compression::Auto;
      if (scan->vcf) {
        exon::VCFConfig cfg = scan->vcf->config();
        cfg.defer_decode = false;
        cfg.threads = 0;
        scan->vcf.reset(new exon::VCFBatchReader(scan->path, c, cfg));
      } else if (scan->fastq) {
        exon::FASTQConfig cfg = scan->fastq->config();
        cfg.defer_decode = false;
        cfg.threads = 0;
        scan->fastq.reset(new exon::FASTQBatchReader(scan->path, c, cfg));
      } else if (scan->bam) {
        exon::BAMConfig cfg = scan->bam->config();
        cfg.threads = 0;
        scan->bam.reset(new exon::BAMBatchReader(scan->path, cfg));
      } else if (scan->bcf) {
        exon::VCFConfig cfg = scan->bcf->config();
        cfg.threads = 0;
        scan->bcf.reset(new exon::BCFBatchReader(scan->path, cfg));
      } else {
        scan->sam.reset(new exon::SAMBatchReader(scan->path, c, scan->sam->config()));
      }
    } catch (const std::exception& e) {
      return fail(nullptr, EXON_HIP_EINVAL, "%s", e.what());
    }
    // fall through to the host paths below
  }
  // K4 grouped by a String / Character INFO key (a dictionary column WITH NULLs from the host readers): the stream turns a NULL
  // key into the id of the empty text in this scan's dictionary -- interned when the first NULL shows up -- so that NULL is a
  // group of its own (DataFusion's GROUP BY); the device parser does the same (exon_hip_vcf_parser_set_null_key)
  struct NullGroupGuard {
    exon_hip_stream* st;
    ~NullGroupGuard() { exon_hip_stream_set_null_group(st, nullptr); }
  } null_group_guard{st};
  if (exon_hip_stream_plan_kind(st) == EXON_HIP_PLAN_CMP_AVG_BY_GROUP && (scan->vcf || scan->bcf)) {
    const std::vector<exon::InfoSpec>& specs = scan->vcf ? scan->vcf->info_specs : scan->bcf->info_specs;
    const int gcol = exon_hip_stream_plan_column(st, 2);
    if (gcol >= 4 && (size_t)(gcol - 4) < specs.size() && specs[(size_t)(gcol - 4)].kind == 's')
      exon_hip_stream_set_null_group(st, [scan, gcol]() -> int32_t {
        exon::Dictionary* d = dict_of(scan, gcol);
        return d ? d->lookup_or_insert("", 0) : -1;
      });
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
        return exon_hip_stream_set_null_group(st, nullptr);  // (flushes what is staged while the scan's dictionary is at hand)
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
  return exon_hip_stream_set_null_group(st, nullptr);  // (flushes what is staged while the scan's dictionary is at hand)
}
