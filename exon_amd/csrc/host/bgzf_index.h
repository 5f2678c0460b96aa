// bgzf_index.h -- tabix (.tbi) / BAI (.bai) binning indexes, region -> BGZF chunk planning, and random-access
// BGZF reading by virtual position.
//
// Reference: exon-core/src/datasources/indexed_file/indexed_bgzf_file.rs:52-155 reads `<file>.tbi` / `<file>.bai`
// fully, calls noodles' `index.query(ref_id, interval)` and turns every returned chunk into one PartitionedFile
// carrying `BGZFIndexedOffsets { start, end }` (virtual positions); the opener then reads the compressed range
// and skips to the intra-block offset (indexed_file_opener.rs:114-162, streaming_bgzf.rs:56-64 does that
// byte-at-a-time; here the block is inflated once and sliced).
// The query is noodles-csi 0.41's BinningIndex::query: bins overlapping the interval (reg2bins, min_shift 14,
// depth 5), their chunks, filtered by the linear index' minimum offset and merged (`optimize_chunks`).
// KAT (same file, :167-187): bigger-index/test.vcf.gz.tbi, chr1:1-3388930 -> one chunk 621346816..3014113427456.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "bgzf_block.h"
#include "io.h"

namespace exon {

struct Chunk {
  uint64_t start = 0, end = 0;  // BGZF virtual positions: (compressed block offset << 16) | offset in block
};

struct RefIndex {
  std::map<uint32_t, std::vector<Chunk>> bins;
  std::vector<uint64_t> linear;  // 16 KiB windows -> smallest virtual offset
};

struct BinningIndex {
  std::vector<std::string> names;  // tabix only (BAI names come from the BAM header)
  std::vector<RefIndex> refs;
  int min_shift = 14, depth = 5;
};

namespace detail {
inline std::vector<uint8_t> slurp(const std::string& path, Compression c) {
  ByteReader r(path, c);
  std::vector<uint8_t> out, buf(1 << 16);
  for (;;) {
    const size_t n = r.read(buf.data(), buf.size());
    if (n == 0) break;
    out.insert(out.end(), buf.begin(), buf.begin() + n);
  }
  return out;
}
struct Cursor {
  const std::vector<uint8_t>& b;
  size_t p = 0;
  template <typename T>
  T get() {
    if (p + sizeof(T) > b.size()) throw std::runtime_error("truncated index");
    T v;
    memcpy(&v, b.data() + p, sizeof(T));
    p += sizeof(T);
    return v;
  }
};
inline void read_refs(Cursor& c, int32_t n_ref, BinningIndex* idx) {
  idx->refs.resize((size_t)n_ref);
  for (int32_t r = 0; r < n_ref; ++r) {
    RefIndex& ri = idx->refs[(size_t)r];
    const int32_t n_bin = c.get<int32_t>();
    for (int32_t b = 0; b < n_bin; ++b) {
      const uint32_t bin = c.get<uint32_t>();
      const int32_t n_chunk = c.get<int32_t>();
      std::vector<Chunk> chunks((size_t)n_chunk);
      for (auto& ch : chunks) {
        ch.start = c.get<uint64_t>();
        ch.end = c.get<uint64_t>();
      }
      if (bin == 37450) continue;  // metadata pseudo-bin
      ri.bins[bin] = std::move(chunks);
    }
    const int32_t n_intv = c.get<int32_t>();
    ri.linear.resize((size_t)n_intv);
    for (auto& v : ri.linear) v = c.get<uint64_t>();
  }
}
}  // namespace detail

inline BinningIndex read_tabix(const std::string& path) {
  const std::vector<uint8_t> raw = detail::slurp(path, Compression::Gzip);  // .tbi is BGZF-compressed
  detail::Cursor c{raw};
  if (raw.size() < 4 || memcmp(raw.data(), "TBI\1", 4) != 0) throw std::runtime_error("not a tabix index: " + path);
  c.p = 4;
  BinningIndex idx;
  const int32_t n_ref = c.get<int32_t>();
  for (int i = 0; i < 6; ++i) c.get<int32_t>();  // format, col_seq, col_beg, col_end, meta, skip
  const int32_t l_nm = c.get<int32_t>();
  if (c.p + (size_t)l_nm > raw.size()) throw std::runtime_error("truncated tabix names");
  size_t s = c.p;
  for (size_t i = c.p; i < c.p + (size_t)l_nm; ++i)
    if (raw[i] == 0) {
      idx.names.emplace_back(reinterpret_cast<const char*>(raw.data() + s), i - s);
      s = i + 1;
    }
  c.p += (size_t)l_nm;
  detail::read_refs(c, n_ref, &idx);
  return idx;
}

inline BinningIndex read_bai(const std::string& path) {
  const std::vector<uint8_t> raw = detail::slurp(path, Compression::None);
  detail::Cursor c{raw};
  if (raw.size() < 4 || memcmp(raw.data(), "BAI\1", 4) != 0) throw std::runtime_error("not a BAI index: " + path);
  c.p = 4;
  BinningIndex idx;
  detail::read_refs(c, c.get<int32_t>(), &idx);
  return idx;
}

// chunks to read for ref `ref_id`, 1-based inclusive [start, end] (end = INT64_MAX: open)
inline std::vector<Chunk> query_index(const BinningIndex& idx, int ref_id, int64_t start, int64_t end) {
  if (ref_id < 0 || ref_id >= (int)idx.refs.size()) return {};
  const RefIndex& ri = idx.refs[(size_t)ref_id];
  const int64_t max_pos = (1ll << (idx.min_shift + 3 * idx.depth)) - 1;
  if (start < 1) start = 1;
  if (end > max_pos) end = max_pos;
  if (start > end) return {};
  // reg2bins over the 0-based half-open interval [start-1, end)
  const int64_t beg = start - 1, e = end - 1;
  std::vector<Chunk> chunks;
  auto take = [&](uint32_t bin) {
    auto it = ri.bins.find(bin);
    if (it != ri.bins.end()) chunks.insert(chunks.end(), it->second.begin(), it->second.end());
  };
  take(0);
  int shift = idx.min_shift + 3 * (idx.depth - 1);
  uint32_t offset = 1;
  for (int level = 1; level <= idx.depth; ++level) {
    for (int64_t k = beg >> shift; k <= (e >> shift); ++k) take(offset + (uint32_t)k);
    offset += 1u << (3 * level);
    shift -= 3;
  }
  // linear index: smallest offset of the 16 KiB window holding `start`
  const size_t win = (size_t)((start - 1) >> idx.min_shift);
  const uint64_t min_offset = win < ri.linear.size() ? ri.linear[win] : 0;
  // optimize_chunks
  std::vector<Chunk> kept;
  for (const auto& c : chunks)
    if (c.end > min_offset) kept.push_back(c);
  if (kept.empty()) return kept;
  std::sort(kept.begin(), kept.end(), [](const Chunk& a, const Chunk& b) { return a.start < b.start; });
  std::vector<Chunk> merged;
  Chunk cur = kept[0];
  for (size_t i = 1; i < kept.size(); ++i) {
    if (kept[i].start > cur.end) {
      merged.push_back(cur);
      cur = kept[i];
    } else if (cur.end < kept[i].end) {
      cur.end = kept[i].end;
    }
  }
  merged.push_back(cur);
  return merged;
}

// Random-access BGZF: inflate whole blocks, track virtual positions.
class BgzfReader {
 public:
  explicit BgzfReader(const std::string& path) : path_(path) {
    f_ = fopen(path.c_str(), "rb");
    if (!f_) throw std::runtime_error("cannot open " + path);
  }
  ~BgzfReader() {
    if (z_init_) inflateEnd(&z_);
    if (f_) fclose(f_);
  }
  BgzfReader(const BgzfReader&) = delete;
  BgzfReader& operator=(const BgzfReader&) = delete;

  void seek(uint64_t vpos) {
    load_block(vpos >> 16);
    pos_ = (size_t)(vpos & 0xFFFF);
    if (pos_ > block_.size()) throw std::runtime_error("virtual offset beyond block: " + path_);
  }
  // virtual position of the next byte to be read
  uint64_t tell() {
    if (pos_ == block_.size() && !eof_) {  // normalise to the start of the next block
      return (uint64_t)(block_off_ + block_csize_) << 16;
    }
    return ((uint64_t)block_off_ << 16) | (uint64_t)pos_;
  }
  bool read_line(std::string* line) {
    line->clear();
    for (;;) {
      if (pos_ == block_.size() && !next_block()) return !line->empty();
      const uint8_t* p = block_.data() + pos_;
      const uint8_t* nl = static_cast<const uint8_t*>(memchr(p, '\n', block_.size() - pos_));
      if (nl) {
        line->append(reinterpret_cast<const char*>(p), nl - p);
        pos_ += (nl - p) + 1;
        if (!line->empty() && line->back() == '\r') line->pop_back();
        return true;
      }
      line->append(reinterpret_cast<const char*>(p), block_.size() - pos_);
      pos_ = block_.size();
    }
  }
  bool read_exact(uint8_t* dst, size_t n) {
    size_t got = 0;
    while (got < n) {
      if (pos_ == block_.size() && !next_block()) return false;
      const size_t k = std::min(n - got, block_.size() - pos_);
      memcpy(dst + got, block_.data() + pos_, k);
      pos_ += k;
      got += k;
    }
    return true;
  }

 private:
  bool next_block() {
    if (eof_) return false;
    do {
      if (!load_block(block_off_ + block_csize_)) return false;
    } while (block_.empty());  // skip empty blocks (EOF marker in the middle of concatenated files)
    pos_ = 0;
    return true;
  }
  bool load_block(int64_t coff) {
    if (fseek(f_, (long)coff, SEEK_SET) != 0) throw std::runtime_error("seek failed: " + path_);
    BgzfBlockInfo info;
    if (!read_bgzf_block(f_, &comp_, &info, path_)) {
      eof_ = true;
      block_.clear();
      pos_ = 0;
      block_off_ = coff;
      block_csize_ = 0;
      return false;
    }
    if (!z_init_) {
      memset(&z_, 0, sizeof z_);
      if (inflateInit2(&z_, -15) != Z_OK) throw std::runtime_error("inflateInit2 failed");
      z_init_ = true;
    }
    block_.resize(info.isize);
    inflate_bgzf_block(&z_, comp_.data(), info, block_.data(), path_);
    block_off_ = coff;
    block_csize_ = (int64_t)info.total;
    eof_ = false;
    return true;
  }
  std::string path_;
  FILE* f_ = nullptr;
  std::vector<uint8_t> block_, comp_;
  z_stream z_;
  bool z_init_ = false;
  size_t pos_ = 0;
  int64_t block_off_ = 0, block_csize_ = 0;
  bool eof_ = false;
};

}  // namespace exon
