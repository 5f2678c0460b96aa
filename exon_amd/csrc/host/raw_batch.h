// raw_batch.h -- internal hand-off between the native decoders and the stream layer: columns as plain vectors
// (values + one validity byte per row), so exon_hip_stream_consume_scan can append decoded slabs straight into the
// pinned staging slot without first materialising an Arrow batch (one copy instead of two on the consumer thread).
#pragma once
#include <cstdint>
#include <vector>

namespace exon {

struct RawColumn {
  const void* values = nullptr;
  const uint8_t* valid_bytes = nullptr;  // one byte per row (1 = valid); nullptr = no nulls
  int elem = 4;                          // bytes per value
};

struct RawBatch {
  int64_t rows = 0;
  std::vector<RawColumn> cols;  // in the scan's column order; valid until the next call on the scan
};

}  // namespace exon
