// scan.cpp -- C ABI of the native decoders (host/formats.h): exon_hip_scan_*.
#include <cstring>
#include <memory>
#include <new>
#include <string>

#include "host/formats.h"
#include "internal.h"

struct exon_hip_scan {
  int format = 0;
  std::unique_ptr<exon::VCFBatchReader> vcf;
  std::unique_ptr<exon::BAMBatchReader> bam;
  std::unique_ptr<exon::SAMBatchReader> sam;
  std::unique_ptr<exon::FASTQBatchReader> fastq;
  std::unique_ptr<exon::FASTABatchReader> fasta;
  int64_t rows = 0;
  exon::Dictionary bam_dict_view;  // reference names as a dictionary (ids = header order)
};

int exon_hip_stream_push_raw(exon_hip_stream* st, const exon::RawBatch& rb);  // stream.cpp

static exon::Dictionary* dict_of(exon_hip_scan* s, int col) {
  if (s->format == EXON_HIP_FORMAT_VCF && col == 0) return &s->vcf->chrom_dict;
  if (s->format == EXON_HIP_FORMAT_VCF && col == 3) return &s->vcf->filter_dict;
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
    else if (s->fastq) s->fastq->schema(out);
    else s->fasta->schema(out);
    return EXON_HIP_OK;
  } catch (const std::exception& e) {
    return fail(nullptr, EXON_HIP_EINVAL, "%s", e.what());
  }
}

int exon_hip_scan_next(exon_hip_scan* s, struct ArrowArray* out) {
  if (!s || !out) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_scan_next: NULL argument");
  try {
    memset(out, 0, sizeof *out);
    bool got;
    if (s->vcf) got = s->vcf->read_batch(out);
    else if (s->bam) got = s->bam->read_batch(out);
    else if (s->sam) got = s->sam->read_batch(out);
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
  delete s;
  return EXON_HIP_OK;
}

int exon_hip_stream_consume_scan(exon_hip_stream* st, exon_hip_scan* scan, int64_t* rows) {
  if (!st || !scan) return fail(nullptr, EXON_HIP_EINVAL, "exon_hip_stream_consume_scan: NULL argument");
  int64_t n = 0;
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
