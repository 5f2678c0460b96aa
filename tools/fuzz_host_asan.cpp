// Sanitizer harness for the host decoders (not part of the library): drains every file given on the command line through one
// reader and counts decoded / rejected inputs.  Build and use (round 2: 3000 corrupted CRAMs, 400-800 corrupted BAM / VCF /
// BCF files each, corrupted before AND after BGZF framing, sequential and threaded readers -- no ASAN / UBSAN finding):
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -Iexon_amd/csrc -Iinclude tools/fuzz_host_asan.cpp -o /tmp/fz -lz -lpthread
//   ASAN_OPTIONS=detect_leaks=0 /tmp/fz {bam|bamt|vcf|vcft|bcf|cram|vcfidx|bamidx} files...   (vcfidx / bamidx: 300 corrupted .tbi / .bai each, no finding)
// Round 3 (lazy CRAM blocks, bzip2 / lzma through dlopen: add -ldl): 600 corrupted CRAMs as built, and 1200 more with
//   -DEXON_CRAM_FUZZ_SKIP_CRC, which lets damaged payloads past the block / container CRC-32 into the decoders -- no finding.
#include "host/cram.h"
#include "host/bcf.h"
#include "host/formats.h"
#include "host/parallel.h"
#include <cstdio>
template <class R> static long drain(R& r) {
  long rows = 0; struct ArrowArray a;
  for (;;) { memset(&a, 0, sizeof a); if (!r.read_batch(&a)) break; rows += a.length; if (a.release) a.release(&a); }
  return rows;
}
int main(int argc, char** argv) {
  long ok = 0, err = 0, rows = 0;
  const std::string kind = argv[1];
  for (int i = 2; i < argc; ++i) {
    try {
      if (kind == "bam") { exon::BAMConfig c; c.threads = 1; exon::BAMBatchReader r(argv[i], c); rows += drain(r); }
      else if (kind == "bamt") { exon::BAMConfig c; c.threads = 4; exon::BAMBatchReader r(argv[i], c); rows += drain(r); }
      else if (kind == "vcf") { exon::VCFConfig c; c.threads = 1; exon::VCFBatchReader r(argv[i], exon::Compression::Auto, c); rows += drain(r); }
      else if (kind == "vcft") { exon::VCFConfig c; c.threads = 4; exon::VCFBatchReader r(argv[i], exon::Compression::Auto, c); rows += drain(r); }
      else if (kind == "vcfidx" || kind == "bamidx") {  // argv[i] = the data file; its (corrupted) .tbi / .bai lies beside it
        exon::RegionFilter rf;
        rf.active = true;
        rf.use_index = true;
        std::string err;
        if (!exon::parse_region(kind == "vcfidx" ? "1:10000000-10000100" : "chr1:1-12209145", &rf.region, &err)) throw std::runtime_error(err);
        if (kind == "vcfidx") { exon::VCFConfig c; c.threads = 1; c.filter = rf; exon::VCFBatchReader r(argv[i], exon::Compression::Auto, c); rows += drain(r); }
        else { exon::BAMConfig c; c.threads = 1; c.filter = rf; exon::BAMBatchReader r(argv[i], c); rows += drain(r); }
      }
      else if (kind == "cram") { exon::BAMConfig c; c.threads = 2; exon::CRAMBatchReader r(argv[i], c); rows += drain(r); }
      else if (kind == "bcf") { exon::VCFConfig c; c.threads = 1; exon::BCFBatchReader r(argv[i], c); rows += drain(r); }
      ++ok;
    } catch (const std::exception& e) { if (err < 2) fprintf(stderr, "%s\n", e.what()); ++err; }
  }
  printf("%s ok %ld err %ld rows %ld\n", kind.c_str(), ok, err, rows);
}
