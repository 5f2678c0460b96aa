// bgzf_block.h -- reading ONE BGZF block from a file with every length checked before it is used.
//
// A BGZF block is a gzip member (RFC 1952) whose extra field holds a "BC" subfield with BSIZE = block size - 1 (SAM
// specification 4.1); the trailer is CRC-32 + ISIZE.  The host readers (sequential, block-parallel and random-access)
// share this parser, so a hostile file cannot make any of them size a buffer from an unchecked field: the block must be
// at least header + trailer long, at most 64 KiB, the BC subfield is FOUND by walking the subfields (it need not be the
// first), ISIZE is bounded by 64 KiB and the CRC-32 of the inflated bytes is verified -- the same checks
// exon_hip_bgzf_scan + the device inflate apply, and noodles bgzf applies in the reference
// (exon-core/src/datasources/vcf/file_opener/unindex_file_opener.rs:62-70 wraps the stream in noodles' bgzf reader).
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace exon {

struct BgzfBlockInfo {
  size_t total = 0;    // bytes of the whole block (BSIZE + 1)
  size_t data = 0;     // offset of the raw DEFLATE data inside the block (12 + XLEN)
  size_t data_len = 0; // total - data - 8
  uint32_t crc32 = 0, isize = 0;
};

constexpr size_t BGZF_MAX_BLOCK = 65536;

// Parses the fixed header + extra field of the block starting at `h` (`avail` readable bytes, >= 12).  Returns the
// number of header bytes needed when `avail` is too short to decide (caller reads more), 0 when *total is known.
inline size_t bgzf_header(const uint8_t* h, size_t avail, size_t* total, size_t* data, const std::string& what) {
  if (avail < 12) return 12;
  if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) throw std::runtime_error("not a BGZF block: " + what);
  const size_t xlen = (size_t)h[10] | ((size_t)h[11] << 8);
  if (avail < 12 + xlen) return 12 + xlen;
  size_t bsize = 0;
  for (size_t x = 12; x + 4 <= 12 + xlen;) {  // subfields: SI1 SI2 SLEN data
    const size_t slen = (size_t)h[x + 2] | ((size_t)h[x + 3] << 8);
    if (x + 4 + slen > 12 + xlen) break;
    if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) bsize = ((size_t)h[x + 4] | ((size_t)h[x + 5] << 8)) + 1;
    x += 4 + slen;
  }
  if (bsize == 0) throw std::runtime_error("BGZF extra field has no BC subfield: " + what);
  if (bsize < 12 + xlen + 8 || bsize > BGZF_MAX_BLOCK) throw std::runtime_error("BGZF block size out of range: " + what);
  *total = bsize;
  *data = 12 + xlen;
  return 0;
}

// Reads the next whole block of `f` into blk (resized to its length).  false at a clean end of file.
inline bool read_bgzf_block(FILE* f, std::vector<uint8_t>* blk, BgzfBlockInfo* info, const std::string& what) {
  blk->resize(12);
  const size_t got = fread(blk->data(), 1, 12, f);
  if (got == 0) return false;
  if (got != 12) throw std::runtime_error("truncated BGZF block: " + what);
  size_t total = 0, data = 0;
  size_t need = bgzf_header(blk->data(), 12, &total, &data, what);
  if (need) {
    blk->resize(need);
    if (fread(blk->data() + 12, 1, need - 12, f) != need - 12) throw std::runtime_error("truncated BGZF block: " + what);
    if (bgzf_header(blk->data(), need, &total, &data, what) != 0) throw std::runtime_error("not a BGZF block: " + what);
  }
  const size_t have = blk->size();
  blk->resize(total);
  if (total > have && fread(blk->data() + have, 1, total - have, f) != total - have) throw std::runtime_error("truncated BGZF block: " + what);
  info->total = total;
  info->data = data;
  info->data_len = total - data - 8;
  memcpy(&info->crc32, blk->data() + total - 8, 4);
  memcpy(&info->isize, blk->data() + total - 4, 4);
  if (info->isize > BGZF_MAX_BLOCK) throw std::runtime_error("BGZF block claims more than 64 KiB: " + what);
  return true;
}

// Same for a block held in memory (`avail` bytes at `p`); throws when the block is not complete.
inline void bgzf_block_info(const uint8_t* p, size_t avail, BgzfBlockInfo* info, const std::string& what) {
  size_t total = 0, data = 0;
  if (bgzf_header(p, avail, &total, &data, what) != 0 || total > avail) throw std::runtime_error("truncated BGZF block: " + what);
  info->total = total;
  info->data = data;
  info->data_len = total - data - 8;
  memcpy(&info->crc32, p + total - 8, 4);
  memcpy(&info->isize, p + total - 4, 4);
  if (info->isize > BGZF_MAX_BLOCK) throw std::runtime_error("BGZF block claims more than 64 KiB: " + what);
}

// Inflates one block into out[0, isize) with a raw-deflate z_stream the caller keeps (inflateInit2(z, -15)); checks
// that the stream ends exactly at ISIZE and that the CRC-32 matches.
inline void inflate_bgzf_block(z_stream* z, const uint8_t* blk, const BgzfBlockInfo& info, uint8_t* out, const std::string& what) {
  if (info.isize) {
    if (inflateReset(z) != Z_OK) throw std::runtime_error("inflateReset failed: " + what);
    z->next_in = const_cast<uint8_t*>(blk + info.data);
    z->avail_in = (uInt)info.data_len;
    z->next_out = out;
    z->avail_out = info.isize;
    if (inflate(z, Z_FINISH) != Z_STREAM_END || z->avail_out != 0) throw std::runtime_error("BGZF inflate error: " + what);
  }
  if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), out, info.isize) != info.crc32) throw std::runtime_error("BGZF CRC-32 mismatch: " + what);
}

}  // namespace exon
