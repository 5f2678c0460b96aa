// io.h -- byte sources for the format readers: plain files and gzip/BGZF (multi-member) via zlib.
//
// Stands where the reference wires `object_store` GET -> `StreamReader` -> optional
// `noodles::bgzf::AsyncReader` / async-compression gzip decoder
// (exon-core/src/datasources/vcf/file_opener/unindex_file_opener.rs:48-92,
//  exon-core/src/datasources/fastq/file_opener.rs:56-105 which sniffs BGZF vs gzip vs plain).
// BGZF is a series of gzip members, so one inflate loop that restarts at member boundaries reads both.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace exon {

enum class Compression { Auto = 0, None = 1, Gzip = 2 };

// a forward-only stream of (decompressed) bytes
class ByteSource {
 public:
  virtual ~ByteSource() = default;
  virtual size_t read(uint8_t* dst, size_t n) = 0;  // up to n bytes; 0 at end of data
  // An uncompressed regular file can also be read with positional reads from several threads: its descriptor and the
  // offset of the next unread byte.  After the caller starts using them it must not call read() again.
  virtual bool plain_file(int* fd, int64_t* offset) { (void)fd; (void)offset; return false; }
};

class ByteReader : public ByteSource {
 public:
  ByteReader(const std::string& path, Compression c) : path_(path) {
    f_ = fopen(path.c_str(), "rb");
    if (!f_) throw std::runtime_error("cannot open " + path);
    in_.resize(1 << 16);
    unsigned char magic[2] = {0, 0};
    size_t got = fread(magic, 1, 2, f_);
    const bool is_gz = got == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    fseek(f_, 0, SEEK_SET);
    gz_ = (c == Compression::Gzip) || (c == Compression::Auto && is_gz);
    if (gz_) {
      memset(&z_, 0, sizeof z_);
      if (inflateInit2(&z_, 15 + 32) != Z_OK) throw std::runtime_error("inflateInit2 failed");
      z_init_ = true;
    }
  }
  ~ByteReader() {
    if (z_init_) inflateEnd(&z_);
    if (f_) fclose(f_);
  }
  bool plain_file(int* fd, int64_t* offset) override {
    if (gz_) return false;
    *fd = fileno(f_);
    *offset = (int64_t)ftell(f_);
    return true;
  }
  ByteReader(const ByteReader&) = delete;
  ByteReader& operator=(const ByteReader&) = delete;

  // up to `n` decompressed bytes; 0 at end of data
  size_t read(uint8_t* dst, size_t n) override {
    if (!gz_) return fread(dst, 1, n, f_);
    size_t produced = 0;
    while (produced < n && !eof_) {
      if (z_.avail_in == 0) {
        z_.avail_in = (uInt)fread(in_.data(), 1, in_.size(), f_);
        z_.next_in = in_.data();
        if (z_.avail_in == 0) {
          // the file ends inside a gzip member (e.g. a header whose XLEN runs past the end): an error, not an empty tail
          if (in_member_) throw std::runtime_error("truncated gzip stream: " + path_);
          eof_ = true;
          break;
        }
      }
      in_member_ = true;
      z_.next_out = dst + produced;
      z_.avail_out = (uInt)(n - produced);
      const int rc = inflate(&z_, Z_NO_FLUSH);
      produced = n - z_.avail_out;
      if (rc == Z_STREAM_END) {
        in_member_ = false;
        // next gzip member (BGZF block) if any bytes remain
        if (z_.avail_in == 0) {
          z_.avail_in = (uInt)fread(in_.data(), 1, in_.size(), f_);
          z_.next_in = in_.data();
        }
        if (z_.avail_in == 0) {
          eof_ = true;
        } else if (inflateReset(&z_) != Z_OK) {
          throw std::runtime_error("inflateReset failed: " + path_);
        }
      } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
        throw std::runtime_error("inflate error in " + path_);
      }
    }
    return produced;
  }

 private:
  std::string path_;
  FILE* f_ = nullptr;
  bool gz_ = false, z_init_ = false, eof_ = false, in_member_ = false;
  z_stream z_;
  std::vector<uint8_t> in_;
};

// buffered line / exact reads over a ByteReader
class BufReader {
 public:
  BufReader(const std::string& path, Compression c) : src_(new ByteReader(path, c)), buf_(1 << 20) {}
  explicit BufReader(std::unique_ptr<ByteSource> src) : src_(std::move(src)), buf_(1 << 20) {}

  // hand the stream over to a parallel decoder: the bytes already buffered but not consumed, then the source
  std::string take_buffered() {
    std::string s(reinterpret_cast<const char*>(buf_.data() + pos_), end_ - pos_);
    pos_ = end_ = 0;
    return s;
  }
  std::unique_ptr<ByteSource> release_source() { return std::move(src_); }
  // bytes handed out so far (position of the next unread byte in the uncompressed stream)
  uint64_t consumed() const { return filled_ - (end_ - pos_); }

  // reads one line without its terminator ('\n' or "\r\n"); false at end of data
  bool read_line(std::string* line) {
    line->clear();
    for (;;) {
      if (pos_ == end_ && !fill()) return !line->empty();
      const uint8_t* p = buf_.data() + pos_;
      const uint8_t* nl = static_cast<const uint8_t*>(memchr(p, '\n', end_ - pos_));
      if (nl) {
        line->append(reinterpret_cast<const char*>(p), nl - p);
        pos_ += (nl - p) + 1;
        if (!line->empty() && line->back() == '\r') line->pop_back();
        return true;
      }
      line->append(reinterpret_cast<const char*>(p), end_ - pos_);
      pos_ = end_;
    }
  }
  // exactly n bytes; false (and a short read) at end of data
  bool read_exact(uint8_t* dst, size_t n) {
    size_t got = 0;
    while (got < n) {
      if (pos_ == end_ && !fill()) return false;
      const size_t k = std::min(n - got, end_ - pos_);
      memcpy(dst + got, buf_.data() + pos_, k);
      pos_ += k;
      got += k;
    }
    return true;
  }
  bool at_end() { return pos_ == end_ && !fill(); }

 private:
  bool fill() {
    end_ = src_ ? src_->read(buf_.data(), buf_.size()) : 0;
    pos_ = 0;
    filled_ += end_;
    return end_ > 0;
  }
  std::unique_ptr<ByteSource> src_;
  std::vector<uint8_t> buf_;
  size_t pos_ = 0, end_ = 0;
  uint64_t filled_ = 0;
};

}  // namespace exon
