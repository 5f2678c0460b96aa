// region.h -- host-side planning helpers that stay on the CPU next to the GPU path.
//
//   parse_region / parse_interval   the noodles_core `Region` / `Interval` grammar the reference parses at
//       exon-core/src/physical_plan/infer_region.rs:31-33 and exon-core/src/udfs/vcf/mod.rs:86-118:
//       `name[:start[-end]]`, 1-based, inclusive, open end allowed; a suffix after the last ':' that is
//       not a valid interval makes the whole string the name.
//   regroup_files_by_size           exon-core/src/datasources/exon_file_scan_config.rs:79-110 -- whole
//       files sorted by ascending size, dealt round-robin into min(target, n) groups.  This is also the
//       rule that assigns file splits to GPUs.
#pragma once
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <string>
#include <vector>

namespace exon {

struct Region {
  std::string name;
  int64_t start = 1;          // 1-based inclusive
  int64_t end = INT64_MAX;    // inclusive; INT64_MAX = open
};

inline bool parse_position(const std::string& s, int64_t* out) {
  if (s.empty() || s.size() > 19) return false;
  size_t i = 0;
  if (s[0] == '+') {
    if (s.size() == 1) return false;
    i = 1;
  }
  int64_t v = 0;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    v = v * 10 + (s[i] - '0');
  }
  if (v < 1) return false;  // positions are non-zero
  *out = v;
  return true;
}

inline bool parse_interval(const std::string& s, int64_t* start, int64_t* end) {
  *start = 1;
  *end = INT64_MAX;
  if (s.empty()) return true;
  const size_t dash = s.find('-');
  if (dash == std::string::npos) return parse_position(s, start);
  return parse_position(s.substr(0, dash), start) && parse_position(s.substr(dash + 1), end);
}

inline bool parse_region(const std::string& s, Region* out, std::string* err) {
  if (s.empty()) {
    if (err) *err = "empty region";
    return false;
  }
  out->name = s;
  out->start = 1;
  out->end = INT64_MAX;
  const size_t colon = s.rfind(':');
  if (colon != std::string::npos) {
    int64_t a, b;
    if (parse_interval(s.substr(colon + 1), &a, &b)) {
      out->name = s.substr(0, colon);
      out->start = a;
      out->end = b;
    }
  }
  if (out->name.empty()) {
    if (err) *err = "empty reference sequence name";
    return false;
  }
  return true;
}

// group_of[i] for every input file i
inline std::vector<int32_t> regroup_files_by_size(const std::vector<int64_t>& sizes, int target_groups) {
  const int n = (int)sizes.size();
  std::vector<int32_t> group_of(n, 0);
  if (n == 0) return group_of;
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sizes[a] < sizes[b]; });
  const int groups = std::max(1, std::min(target_groups, n));
  for (int i = 0; i < n; ++i) group_of[order[i]] = i % groups;
  return group_of;
}

}  // namespace exon
