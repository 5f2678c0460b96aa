// text_columns.hip -- the reference's STRING / LIST columns out of the GPU decode pipeline: Arrow Utf8 / List<Utf8> / List<Int64>
// buffers (offsets + compact values) built on the device from the slab the parsers have just indexed.
//
// Reference columns rebuilt here (the ones outside the fused kernels' operands, selected by exon_hip_scan_options.projection):
//   VCF  id         List<Utf8>, NULL when the ID field is '.'       exon-vcf/src/array_builder/lazy_array_builder.rs:169-180
//        ref        Utf8                                             :181-190
//        alt        List<Utf8>: NULL when ALT is '.', otherwise a list with NO items -- the reference concatenates the alternate
//                   bases into a local string and then calls `alternates.append(true)` without ever appending a value (:191-205);
//                   reproduced as it is (results identical to the reference's), the quirk is written down in DESIGN.md
//   BAM  name       Utf8, NULL for '*'                                exon-bam/src/array_builder.rs:105-113
//        cigar      Utf8, "<len><op>..." with ops MIDNSHP=X           :144-167
//        sequence   Utf8, 4-bit codes through "=ACMGRSVTWYHKDBN"      :178-183
//        quality_scores  List<Int64>, the raw bytes as i8 -> i64      :184-201
// One thread per row measures, an exclusive scan turns lengths into offsets, one thread per row fills (rows are short; the bytes
// of a slab are read twice, L2-resident the second time).  Nothing here is on the fused kernels' path: a scan that projects none of
// these columns launches none of this.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>

#include "internal.h"

namespace {

constexpr int TPB = 256;

// ---- exclusive scan of n u32 lengths into n + 1 int32 offsets (three launches; n is known on the host) ------------------------------
__global__ __launch_bounds__(TPB) void k_block_sums(const uint32_t* __restrict__ len, unsigned n, unsigned* __restrict__ sums) {
  __shared__ unsigned red[TPB / 64];
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  unsigned c = i < n ? len[i] : 0u;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void k_scan_sums(unsigned* __restrict__ sums, int nb, unsigned* __restrict__ total) {
  __shared__ unsigned part[256];
  const int per = (nb + 255) / 256;
  const int b0 = threadIdx.x * per, b1 = min(nb, b0 + per);
  unsigned s = 0;
  for (int b = b0; b < b1; ++b) s += sums[b];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {
    const unsigned v = threadIdx.x >= (unsigned)o ? part[threadIdx.x - o] : 0u;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  unsigned run = threadIdx.x ? part[threadIdx.x - 1] : 0u;
  for (int b = b0; b < b1; ++b) {
    const unsigned c = sums[b];
    sums[b] = run;
    run += c;
  }
  if (threadIdx.x == 255) *total = part[255];
}
__global__ __launch_bounds__(TPB) void k_write_offsets(const uint32_t* __restrict__ len, unsigned n, const unsigned* __restrict__ sums, int32_t* __restrict__ offsets) {
  __shared__ unsigned wave_tot[TPB / 64];
  const unsigned i = blockIdx.x * TPB + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned c = i < n ? len[i] : 0u;
  unsigned incl = c;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_tot[wave] = incl;
  __syncthreads();
  unsigned base = sums[blockIdx.x];
  for (int w = 0; w < wave; ++w) base += wave_tot[w];
  if (i < n) offsets[i] = (int32_t)(base + incl - c);
  if (i == n - 1) offsets[n] = (int32_t)(base + incl);
}

// ---- VCF ---------------------------------------------------------------------------------------------------------------------------
// line r of the slab: [begin, end) in the aligned text; fields 2, 3, 4 (ID, REF, ALT) by their tabs
struct VcfLens {
  uint32_t* id_items;   // items of the ID list (0 for '.')
  uint32_t* id_bytes;   // bytes of its items (the ';' between them do not count)
  uint32_t* ref_bytes;
  uint32_t* field_off;  // [3 n]: where ID, REF, ALT start
  uint32_t* field_len;  // [3 n]
};
__global__ __launch_bounds__(TPB) void k_vcf_measure(const uint8_t* __restrict__ text, const unsigned* __restrict__ nl, unsigned n_rows, unsigned skip, VcfLens o,
                                                     uint32_t* __restrict__ id_valid, uint32_t* __restrict__ alt_valid) {
  const unsigned row = blockIdx.x * TPB + threadIdx.x;
  bool idv = false, altv = false;
  if (row < n_rows) {
    const unsigned begin = row ? nl[row - 1] + 1 : skip;
    unsigned end = nl[row];
    if (end > begin && text[end - 1] == '\r') --end;
    unsigned fs[6];
    int nf = 0;
    fs[0] = begin;
    for (unsigned i = begin; i < end && nf < 5; ++i)
      if (text[i] == '\t') fs[++nf] = i + 1;
    // (rows with fewer than 5 fields are not data lines: the parser has counted them undecided, the slab goes to the host reader)
    unsigned off[3] = {0, 0, 0}, len[3] = {0, 0, 0};
    for (int f = 0; f < 3; ++f)
      if (nf >= f + 3) {
        off[f] = fs[f + 2];
        len[f] = fs[f + 3] - 1 - fs[f + 2];
      } else if (nf == f + 2) {
        off[f] = fs[f + 2];
        len[f] = end - fs[f + 2];
      }
    unsigned items = 0, bytes = 0;
    if (!(len[0] == 0 || (len[0] == 1 && text[off[0]] == '.'))) {
      items = 1;
      bytes = len[0];
      for (unsigned i = 0; i < len[0]; ++i)
        if (text[off[0] + i] == ';') {
          ++items;
          --bytes;
        }
      idv = true;
    }
    altv = !(len[2] == 0 || (len[2] == 1 && text[off[2]] == '.'));
    o.id_items[row] = items;
    o.id_bytes[row] = bytes;
    o.ref_bytes[row] = len[1];
    for (int f = 0; f < 3; ++f) {
      o.field_off[3 * row + f] = off[f];
      o.field_len[3 * row + f] = len[f];
    }
  }
  // validity bitmaps: one word per 32 rows (a wave covers two)
  const unsigned long long bi = __ballot(idv), ba = __ballot(altv);
  const unsigned lane = threadIdx.x & 63u, wrow = row - lane;
  if (lane < 2 && wrow + 32 * lane < n_rows) {
    id_valid[(wrow >> 5) + lane] = (uint32_t)(bi >> (32 * lane));
    alt_valid[(wrow >> 5) + lane] = (uint32_t)(ba >> (32 * lane));
  }
}
__global__ __launch_bounds__(TPB) void k_vcf_fill(const uint8_t* __restrict__ text, unsigned n_rows, VcfLens o, const int32_t* __restrict__ id_list_off,
                                                  const int32_t* __restrict__ id_byte_off, const int32_t* __restrict__ ref_off, int32_t* __restrict__ id_item_off,
                                                  uint8_t* __restrict__ id_values, uint8_t* __restrict__ ref_values, unsigned id_items_total, unsigned id_bytes_total) {
  const unsigned row = blockIdx.x * TPB + threadIdx.x;
  if (row >= n_rows) return;
  {  // REF
    const unsigned off = o.field_off[3 * row + 1], len = o.field_len[3 * row + 1];
    uint8_t* dst = ref_values + ref_off[row];
    for (unsigned i = 0; i < len; ++i) dst[i] = text[off + i];
  }
  const unsigned items = o.id_items[row];
  if (items) {  // ID: the items back to back, an offset per item
    const unsigned off = o.field_off[3 * row + 0], len = o.field_len[3 * row + 0];
    unsigned k = (unsigned)id_list_off[row], w = (unsigned)id_byte_off[row];
    id_item_off[k++] = (int32_t)w;
    for (unsigned i = 0; i < len; ++i) {
      const uint8_t c = text[off + i];
      if (c == ';') id_item_off[k++] = (int32_t)w;
      else id_values[w++] = c;
    }
  }
  if (row == n_rows - 1) id_item_off[id_items_total] = (int32_t)id_bytes_total;
}

// ---- BAM ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
__device__ __forceinline__ unsigned dec_digits(uint32_t v) {
  unsigned d = 1;
  while (v >= 10) {
    v /= 10;
    ++d;
  }
  return d;
}
struct BamLens {
  uint32_t* name_bytes;
  uint32_t* cigar_bytes;
  uint32_t* seq_bytes;  // = items of quality_scores
};
__global__ __launch_bounds__(TPB) void k_bam_measure(const uint8_t* __restrict__ d, const uint32_t* __restrict__ rec_of_row, unsigned n_rows, BamLens o,
                                                     uint32_t* __restrict__ name_valid) {
  const unsigned row = blockIdx.x * TPB + threadIdx.x;
  bool nv = false;
  if (row < n_rows) {
    const uint8_t* r = d + rec_of_row[row];
    const uint32_t l_name = r[12], n_cigar = (uint32_t)r[16] | (uint32_t)r[17] << 8, l_seq = ld32u(r + 20);
    // noodles: a read name of "*" is a missing name (Record::name -> None)
    nv = !(l_name == 2 && r[36] == '*');
    o.name_bytes[row] = nv ? l_name - 1 : 0u;
    unsigned cb = 0;
    const uint8_t* c = r + 36 + l_name;
    for (uint32_t k = 0; k < n_cigar; ++k) cb += dec_digits(ld32u(c + 4 * k) >> 4) + 1;
    o.cigar_bytes[row] = cb;
    o.seq_bytes[row] = l_seq;
  }
  const unsigned long long b = __ballot(nv);
  const unsigned lane = threadIdx.x & 63u, wrow = row - lane;
  if (lane < 2 && wrow + 32 * lane < n_rows) name_valid[(wrow >> 5) + lane] = (uint32_t)(b >> (32 * lane));
}
__global__ __launch_bounds__(TPB) void k_bam_fill(const uint8_t* __restrict__ d, const uint32_t* __restrict__ rec_of_row, unsigned n_rows, uint64_t projection,
                                                  const int32_t* __restrict__ name_off, const int32_t* __restrict__ cigar_off, const int32_t* __restrict__ seq_off,
                                                  uint8_t* __restrict__ name_values, uint8_t* __restrict__ cigar_values, uint8_t* __restrict__ seq_values,
                                                  int64_t* __restrict__ qual_values) {
  const unsigned row = blockIdx.x * TPB + threadIdx.x;
  if (row >= n_rows) return;
  const uint8_t* r = d + rec_of_row[row];
  const uint32_t l_name = r[12], n_cigar = (uint32_t)r[16] | (uint32_t)r[17] << 8, l_seq = ld32u(r + 20);
  if (projection & EXON_HIP_PROJECT_BAM_NAME) {
    const unsigned n = (unsigned)(name_off[row + 1] - name_off[row]);
    uint8_t* dst = name_values + name_off[row];
    for (unsigned i = 0; i < n; ++i) dst[i] = r[36 + i];
  }
  const uint8_t* c = r + 36 + l_name;
  if (projection & EXON_HIP_PROJECT_BAM_CIGAR) {
    uint8_t* dst = cigar_values + cigar_off[row];
    for (uint32_t k = 0; k < n_cigar; ++k) {
      const uint32_t op = ld32u(c + 4 * k);
      uint32_t v = op >> 4;
      const unsigned nd = dec_digits(v);
      for (unsigned i = nd; i > 0; --i) {
        dst[i - 1] = (uint8_t)('0' + v % 10);
        v /= 10;
      }
      dst += nd;
      const uint32_t code = op & 0xF;
      *dst++ = code < 9 ? (uint8_t)"MIDNSHP=X"[code] : (uint8_t)'?';
    }
  }
  const uint8_t* s = c + 4 * n_cigar;
  if (projection & EXON_HIP_PROJECT_BAM_SEQUENCE) {
    uint8_t* dst = seq_values + seq_off[row];
    for (uint32_t i = 0; i < l_seq; ++i) dst[i] = (uint8_t)"=ACMGRSVTWYHKDBN"[(s[i >> 1] >> ((i & 1) ? 0 : 4)) & 0xF];
  }
  if (projection & EXON_HIP_PROJECT_BAM_QUALITY_SCORES) {
    const uint8_t* q = s + (l_seq + 1) / 2;
    int64_t* dst = qual_values + seq_off[row];
    for (uint32_t i = 0; i < l_seq; ++i) dst[i] = (int64_t)(int8_t)q[i];
  }
}

}  // namespace

// ---- host side -----------------------------------------------------------------------------------------------------------------------
// ---- FASTQ -------------------------------------------------------------------------------------------------------------------------
// name, description, sequence, quality_scores (exon-fastq/src/array_builder.rs:68-102; schema exon-fastq/src/config.rs:79-88): the
// header line behind '@' up to the first space is the name, what follows the space the description (NULL when there is no space
// or nothing behind it); sequence and quality lines as they are (CR dropped by the views).
struct FastqLens {
  uint32_t *name, *desc, *seq, *qual;
};
__global__ __launch_bounds__(TPB) void k_fastq_measure(const uint8_t* __restrict__ text, unsigned n_reads, const int32_t* __restrict__ head_s,
                                                       const int32_t* __restrict__ head_e, const int32_t* __restrict__ seq_s, const int32_t* __restrict__ seq_e,
                                                       const int32_t* __restrict__ qual_s, const int32_t* __restrict__ qual_e, FastqLens o,
                                                       uint32_t* __restrict__ desc_valid) {
  const unsigned r = blockIdx.x * TPB + threadIdx.x;
  bool has_desc = false;
  if (r < n_reads) {
    const unsigned b = (unsigned)head_s[r], e = (unsigned)head_e[r];
    unsigned sp = b;
    while (sp < e && text[sp] != ' ') ++sp;
    o.name[r] = sp - b;
    has_desc = sp + 1 < e;
    o.desc[r] = has_desc ? e - sp - 1 : 0u;
    o.seq[r] = (unsigned)(seq_e[r] - seq_s[r]);
    o.qual[r] = (unsigned)(qual_e[r] - qual_s[r]);
  }
  const unsigned long long m = __ballot(has_desc);
  if ((threadIdx.x & 63) == 0 && r < n_reads + 63) {
    desc_valid[r >> 5] = (uint32_t)m;
    desc_valid[(r >> 5) + 1] = (uint32_t)(m >> 32);
  }
}
// bytes [src, src + n) of the text to dst: 8 at a time (unaligned on both sides: the hardware takes it), the rest one by one
__device__ __forceinline__ void copy_run(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, unsigned n) {
  unsigned i = 0;
  for (; i + 8 <= n; i += 8) {
    const uint32_t a = *reinterpret_cast<const uint32_t*>(src + i), b = *reinterpret_cast<const uint32_t*>(src + i + 4);
    *reinterpret_cast<uint32_t*>(dst + i) = a;
    *reinterpret_cast<uint32_t*>(dst + i + 4) = b;
  }
  for (; i < n; ++i) dst[i] = src[i];
}
__global__ __launch_bounds__(TPB) void k_fastq_fill(const uint8_t* __restrict__ text, unsigned n_reads, const int32_t* __restrict__ head_s,
                                                    const int32_t* __restrict__ seq_s, const int32_t* __restrict__ qual_s, const int32_t* __restrict__ name_off,
                                                    const int32_t* __restrict__ desc_off, const int32_t* __restrict__ seq_off, const int32_t* __restrict__ qual_off,
                                                    uint8_t* __restrict__ name_v, uint8_t* __restrict__ desc_v, uint8_t* __restrict__ seq_v,
                                                    uint8_t* __restrict__ qual_v) {
  const unsigned r = blockIdx.x * TPB + threadIdx.x;
  if (r >= n_reads) return;
  const unsigned nn = (unsigned)(name_off[r + 1] - name_off[r]), nd = (unsigned)(desc_off[r + 1] - desc_off[r]);
  copy_run(name_v + name_off[r], text + head_s[r], nn);
  if (nd) copy_run(desc_v + desc_off[r], text + head_s[r] + nn + 1, nd);
  copy_run(seq_v + seq_off[r], text + seq_s[r], (unsigned)(seq_off[r + 1] - seq_off[r]));
  copy_run(qual_v + qual_off[r], text + qual_s[r], (unsigned)(qual_off[r + 1] - qual_off[r]));
}

// ---- SAM ---------------------------------------------------------------------------------------------------------------------------
// The BAM columns from an alignment LINE (exon-sam/src/array_builder.rs:101-185 over noodles' RecordBuf): QNAME '*' -> NULL; the
// CIGAR printed op by op ('*' -> ""; a text the printer would change -- an op count with a leading zero -- or would refuse makes
// the row undecided: the host reader takes the file); SEQ '*' -> ""; QUAL '*' -> an empty list, else Phred = char - 33 (a
// character outside '!'..'~' is the reader's error: undecided).  Fields 1 (QNAME), 6 (CIGAR), 10 (SEQ), 11 (QUAL) by their tabs.
struct SamLens {
  uint32_t *name, *cigar, *seq, *qual;
  uint32_t* field_off;  // [4 n]: where QNAME, CIGAR, SEQ, QUAL start
};
__global__ __launch_bounds__(TPB) void k_sam_measure(const uint8_t* __restrict__ text, const unsigned* __restrict__ nl, unsigned n_rows, unsigned skip, SamLens o,
                                                     uint32_t* __restrict__ name_valid, unsigned* __restrict__ undecided) {
  const unsigned row = blockIdx.x * TPB + threadIdx.x;
  bool nv = false, bad = false;
  if (row < n_rows) {
    const unsigned begin = row ? nl[row - 1] + 1 : skip;
    unsigned end = nl[row];
    if (end > begin && text[end - 1] == '\r') --end;
    unsigned fs[12];
    int nf = 0;
    fs[0] = begin;
    for (unsigned i = begin; i < end && nf < 11; ++i)
      if (text[i] == '\t') fs[++nf] = i + 1;
    unsigned off[4] = {begin, begin, begin, begin}, len[4] = {0, 0, 0, 0};
    if (nf < 10) {
      bad = true;  // fewer than 11 fields
    } else {
      const int which[4] = {0, 5, 9, 10};
      for (int k = 0; k < 4; ++k) {
        const int f = which[k];
        off[k] = fs[f];
        len[k] = (f < nf ? fs[f + 1] - 1 : end) - fs[f];
      }
      auto star = [&](int k) { return len[k] == 1 && text[off[k]] == '*'; };
      nv = !star(0);
      if (star(0)) len[0] = 0;
      if (star(1)) {
        len[1] = 0;
      } else {  // digits (no leading zero: the printed form is the text) then one of MIDNSHP=X, repeated
        bool digits = false, lead = true;
        for (unsigned i = 0; i < len[1]; ++i) {
          const uint8_t c = text[off[1] + i];
          if (c >= '0' && c <= '9') {
            if (lead && c == '0') bad = true;
            lead = false;
            digits = true;
          } else {
            const bool op = c == 'M' || c == 'I' || c == 'D' || c == 'N' || c == 'S' || c == 'H' || c == 'P' || c == '=' || c == 'X';
            if (!digits || !op) bad = true;
            digits = false;
            lead = true;
          }
        }
        if (digits || len[1] == 0) bad = true;
      }
      if (star(2)) len[2] = 0;
      if (star(3)) {
        len[3] = 0;
      } else {
        for (unsigned i = 0; i < len[3]; ++i) {
          const uint8_t c = text[off[3] + i];
          if (c < 33 || c > 126) bad = true;
        }
      }
    }
    o.name[row] = len[0];
    o.cigar[row] = len[1];
    o.seq[row] = len[2];
    o.qual[row] = len[3];
    for (int k = 0; k < 4; ++k) o.field_off[4 * row + k] = off[k];
  }
  const unsigned long long bv = __ballot(nv), bb = __ballot(bad);
  const unsigned lane = threadIdx.x & 63u, wrow = row - lane;
  if (lane < 2 && wrow + 32 * lane < n_rows) name_valid[(wrow >> 5) + lane] = (uint32_t)(bv >> (32 * lane));
  if (lane == 0 && bb) atomicAdd(undecided, (unsigned)__popcll(bb));
}
__global__ __launch_bounds__(TPB) void k_sam_fill(const uint8_t* __restrict__ text, unsigned n_rows, SamLens o, uint64_t projection, const int32_t* __restrict__ name_off,
                                                  const int32_t* __restrict__ cigar_off, const int32_t* __restrict__ seq_off, const int32_t* __restrict__ qual_off,
                                                  uint8_t* __restrict__ name_v, uint8_t* __restrict__ cigar_v, uint8_t* __restrict__ seq_v, int64_t* __restrict__ qual_v) {
  const unsigned row = blockIdx.x * TPB + threadIdx.x;
  if (row >= n_rows) return;
  const uint32_t* fo = o.field_off + 4 * row;
  if (projection & EXON_HIP_PROJECT_BAM_NAME) copy_run(name_v + name_off[row], text + fo[0], o.name[row]);
  if (projection & EXON_HIP_PROJECT_BAM_CIGAR) copy_run(cigar_v + cigar_off[row], text + fo[1], o.cigar[row]);
  if (projection & EXON_HIP_PROJECT_BAM_SEQUENCE) copy_run(seq_v + seq_off[row], text + fo[2], o.seq[row]);
  if (projection & EXON_HIP_PROJECT_BAM_QUALITY_SCORES) {
    int64_t* dst = qual_v + qual_off[row];
    const uint8_t* q = text + fo[3];
    for (unsigned i = 0; i < o.qual[row]; ++i) dst[i] = (int64_t)q[i] - 33;
  }
}

// ---- BCF ---------------------------------------------------------------------------------------------------------------------------
// id / ref / alt of a BCF record (VCF specification 6.3.1): behind the 24 fixed bytes of the shared block (the record starts with
// l_shared, l_indiv) the ID as a typed string (';'-separated, "." = none) and n_allele typed strings, the first one REF.  A typed
// value's descriptor byte is count << 4 | type (7 = characters); count 15 = the real count follows as a typed integer.
struct BcfLens {
  uint32_t *id_items, *id_bytes, *ref_bytes, *alt_items, *alt_bytes;
};
// the typed string at p: where its characters start and how many there are; false: not a string (or past `end`)
__device__ __forceinline__ bool bcf_typed_string(const uint8_t* d, uint32_t* p, uint32_t end, uint32_t* at, uint32_t* len) {
  if (*p >= end) return false;
  const uint8_t b = d[(*p)++];
  uint32_t n = b >> 4;
  const uint32_t t = b & 15u;
  if (n == 15) {
    if (*p >= end) return false;
    const uint8_t c = d[(*p)++];
    const uint32_t ct = c & 15u;
    if ((c >> 4) != 1 || ct < 1 || ct > 3) return false;
    const uint32_t w = ct == 1 ? 1u : ct == 2 ? 2u : 4u;
    if (*p + w > end) return false;
    n = 0;
    for (uint32_t i = 0; i < w; ++i) n |= (uint32_t)d[*p + i] << (8 * i);
    *p += w;
  }
  if (!(t == 7 || (t == 0 && n == 0))) return false;
  if (*p + n > end) return false;
  *at = *p;
  *len = t == 7 ? n : 0u;
  *p += *len;
  return true;
}
__global__ __launch_bounds__(TPB) void k_bcf_measure(const uint8_t* __restrict__ d, const uint32_t* __restrict__ rec_of_row, unsigned n_rows, BcfLens o,
                                                     unsigned* __restrict__ undecided) {
  const unsigned row = blockIdx.x * TPB + threadIdx.x;
  bool bad = false;
  if (row < n_rows) {
    const uint32_t r = rec_of_row[row];
    const uint32_t ls = ld32u(d + r), nia = ld32u(d + r + 24);
    const uint32_t n_allele = nia >> 16, end = r + 8 + ls;
    uint32_t p = r + 32, at = 0, len = 0;
    unsigned id_items = 0, id_bytes = 0, ref_bytes = 0, alt_items = 0, alt_bytes = 0;
    if (!bcf_typed_string(d, &p, end, &at, &len)) bad = true;
    if (!bad && !(len == 0 || (len == 1 && d[at] == '.'))) {
      id_items = 1;
      id_bytes = len;
      for (uint32_t i = 0; i < len; ++i)
        if (d[at + i] == ';') {
          ++id_items;
          --id_bytes;
        }
    }
    for (uint32_t a = 0; a < n_allele && !bad; ++a) {
      if (!bcf_typed_string(d, &p, end, &at, &len)) {
        bad = true;
        break;
      }
      if (a == 0) ref_bytes = len;
      else {
        ++alt_items;
        alt_bytes += len;
      }
    }
    o.id_items[row] = bad ? 0u : id_items;
    o.id_bytes[row] = bad ? 0u : id_bytes;
    o.ref_bytes[row] = bad ? 0u : ref_bytes;
    o.alt_items[row] = bad ? 0u : alt_items;
    o.alt_bytes[row] = bad ? 0u : alt_bytes;
  }
  const unsigned long long bb = __ballot(bad);
  if ((threadIdx.x & 63) == 0 && bb) atomicAdd(undecided, (unsigned)__popcll(bb));
}
__global__ __launch_bounds__(TPB) void k_bcf_fill(const uint8_t* __restrict__ d, const uint32_t* __restrict__ rec_of_row, unsigned n_rows, uint64_t projection,
                                                  const int32_t* __restrict__ id_list_off, const int32_t* __restrict__ id_byte_off, const int32_t* __restrict__ ref_off,
                                                  const int32_t* __restrict__ alt_list_off, const int32_t* __restrict__ alt_byte_off, int32_t* __restrict__ id_item_off,
                                                  int32_t* __restrict__ alt_item_off, uint8_t* __restrict__ id_values, uint8_t* __restrict__ ref_values,
                                                  uint8_t* __restrict__ alt_values, unsigned id_items_total, unsigned id_bytes_total, unsigned alt_items_total,
                                                  unsigned alt_bytes_total) {
  const unsigned row = blockIdx.x * TPB + threadIdx.x;
  if (row >= n_rows) return;
  const uint32_t r = rec_of_row[row];
  const uint32_t ls = ld32u(d + r), nia = ld32u(d + r + 24);
  const uint32_t n_allele = nia >> 16, end = r + 8 + ls;
  uint32_t p = r + 32, at = 0, len = 0;
  if (bcf_typed_string(d, &p, end, &at, &len) && (projection & EXON_HIP_PROJECT_VCF_ID) && id_list_off[row + 1] > id_list_off[row]) {
    unsigned k = (unsigned)id_list_off[row], w = (unsigned)id_byte_off[row];
    id_item_off[k++] = (int32_t)w;
    for (uint32_t i = 0; i < len; ++i) {
      const uint8_t c = d[at + i];
      if (c == ';') id_item_off[k++] = (int32_t)w;
      else id_values[w++] = c;
    }
  }
  unsigned ak = (unsigned)alt_list_off[row], aw = (unsigned)alt_byte_off[row];
  for (uint32_t a = 0; a < n_allele; ++a) {
    if (!bcf_typed_string(d, &p, end, &at, &len)) break;
    if (a == 0) {
      if (projection & EXON_HIP_PROJECT_VCF_REF) copy_run(ref_values + ref_off[row], d + at, len);
    } else if (projection & EXON_HIP_PROJECT_VCF_ALT) {
      alt_item_off[ak++] = (int32_t)aw;
      copy_run(alt_values + aw, d + at, len);
      aw += len;
    }
  }
  if (row == n_rows - 1) {
    id_item_off[id_items_total] = (int32_t)id_bytes_total;
    alt_item_off[alt_items_total] = (int32_t)alt_bytes_total;
  }
}

struct ExonTextScratch {
  exon_hip_ctx* ctx = nullptr;
  int64_t max_rows = 0, max_bytes = 0;
  int n_cols = 3;
  uint32_t* len[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  int32_t* off[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  uint32_t* valid[2] = {nullptr, nullptr};
  uint32_t *field_off = nullptr, *field_len = nullptr;
  uint32_t* sam_field_off = nullptr;  // [4 r] (SAM)
  unsigned* sums = nullptr;
  unsigned* totals = nullptr;    // device [4]
  unsigned* h_totals = nullptr;  // pinned
  int32_t* item_off = nullptr;   // VCF / BCF id items
  int32_t* item_off2 = nullptr;  // BCF alt items
  unsigned* totals5 = nullptr;   // device [8]: the five totals of the BCF columns
  unsigned* h_totals5 = nullptr; // pinned
  uint8_t* values[4] = {nullptr, nullptr, nullptr, nullptr};
  int64_t* qual = nullptr;
  size_t qual_cap = 0;
};

void exon_text_scratch_destroy(ExonTextScratch* s) {
  if (!s) return;
  auto f = [&](void* p) { if (p) exon_pool_free(s->ctx, p); };
  for (int k = 0; k < 4; ++k) f(s->values[k]);
  for (int k = 0; k < 5; ++k) f(s->len[k]), f(s->off[k]);
  f(s->item_off2), f(s->totals5);
  if (s->h_totals5) hipHostFree(s->h_totals5);
  f(s->valid[0]), f(s->valid[1]), f(s->field_off), f(s->field_len), f(s->sam_field_off), f(s->sums), f(s->totals), f(s->item_off), f(s->qual);
  if (s->h_totals) hipHostFree(s->h_totals);
  delete s;
}

static int scratch_for(exon_hip_ctx* ctx, ExonTextScratch** sp, int64_t max_rows, int64_t max_bytes, bool vcf, int n_cols = 3) {
  ExonTextScratch* s = *sp;
  if (s && s->max_rows >= max_rows && s->max_bytes >= max_bytes && s->n_cols >= n_cols) return EXON_HIP_OK;
  if (s) exon_text_scratch_destroy(s);
  *sp = nullptr;
  s = new (std::nothrow) ExonTextScratch();
  if (!s) return fail(ctx, EXON_HIP_ENOMEM, "out of host memory");
  s->ctx = ctx;
  s->max_rows = max_rows;
  s->max_bytes = max_bytes;
  s->n_cols = n_cols;
  hipSetDevice(ctx->device);
  bool ok = true;
  auto a = [&](void** p, size_t bytes) {
    if (ok && !(*p = exon_pool_alloc(ctx, bytes))) ok = false;
  };
  const size_t r = (size_t)max_rows + 64;
  for (int k = 0; k < n_cols; ++k) {
    a((void**)&s->len[k], r * 4);
    a((void**)&s->off[k], (r + 1) * 4);
    if (k < 4) a((void**)&s->values[k], (size_t)max_bytes + 64);
  }
  if (n_cols >= 5) {  // BCF: alt items
    a((void**)&s->item_off2, ((size_t)max_bytes / 2 + r + 2) * 4);
    a((void**)&s->totals5, 32);
    if (ok && hipHostMalloc((void**)&s->h_totals5, 32) != hipSuccess) ok = false;
  }
  a((void**)&s->valid[0], r / 8 + 64);
  a((void**)&s->valid[1], r / 8 + 64);
  if (vcf) {
    a((void**)&s->field_off, 3 * r * 4);
    a((void**)&s->field_len, 3 * r * 4);
    a((void**)&s->item_off, ((size_t)max_bytes / 2 + r + 2) * 4);
  }
  if (n_cols >= 4) a((void**)&s->sam_field_off, 4 * r * 4);
  a((void**)&s->sums, (r / TPB + 4) * 4);
  a((void**)&s->totals, 16);
  if (ok && hipHostMalloc((void**)&s->h_totals, 16) != hipSuccess) ok = false;  // (4 totals: the FASTQ columns use them all)
  if (!ok) {
    (void)hipGetLastError();
    exon_text_scratch_destroy(s);
    return fail(ctx, EXON_HIP_ENOMEM, "buffers for the string columns of a slab (%lld rows, %lld bytes)", (long long)max_rows, (long long)max_bytes);
  }
  *sp = s;
  return EXON_HIP_OK;
}

static void scan_lengths(hipStream_t hs, ExonTextScratch* s, const uint32_t* len, unsigned n, int32_t* offsets, int total_slot) {
  const int nb = (int)((n + TPB - 1) / TPB);
  hipLaunchKernelGGL(k_block_sums, dim3(nb), dim3(TPB), 0, hs, len, n, s->sums);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, hs, s->sums, nb, s->totals + total_slot);
  hipLaunchKernelGGL(k_write_offsets, dim3(nb), dim3(TPB), 0, hs, len, n, s->sums, offsets);
}

int exon_text_vcf(exon_hip_ctx* ctx, void* stream, ExonTextScratch** sp, const uint8_t* d_text, int64_t n_bytes, const unsigned* d_nl, int64_t n_rows, uint64_t projection,
                  ExonVcfText* out) {
  memset(out, 0, sizeof *out);
  if (n_rows == 0 || !(projection & (EXON_HIP_PROJECT_VCF_ID | EXON_HIP_PROJECT_VCF_REF | EXON_HIP_PROJECT_VCF_ALT))) return EXON_HIP_OK;
  const unsigned skip = (unsigned)(reinterpret_cast<uintptr_t>(d_text) & 15);
  d_text -= skip;
  n_bytes += skip;
  int rc = scratch_for(ctx, sp, std::max<int64_t>(n_rows, 1 << 16), std::max<int64_t>(n_bytes, 1 << 20), true);
  if (rc) return rc;
  ExonTextScratch* s = *sp;
  hipStream_t hs = pick_stream(ctx, stream);
  const unsigned n = (unsigned)n_rows;
  const int nb = (int)((n + TPB - 1) / TPB);
  VcfLens L{s->len[0], s->len[1], s->len[2], s->field_off, s->field_len};
  hipLaunchKernelGGL(k_vcf_measure, dim3(nb), dim3(TPB), 0, hs, d_text, d_nl, n, skip, L, s->valid[0], s->valid[1]);
  scan_lengths(hs, s, s->len[0], n, s->off[0], 0);  // ID: list offsets
  scan_lengths(hs, s, s->len[1], n, s->off[1], 1);  // ID: byte offsets of every row's items
  scan_lengths(hs, s, s->len[2], n, s->off[2], 2);  // REF
  HIP_TRY(ctx, hipMemcpyAsync(s->h_totals, s->totals, 16, hipMemcpyDeviceToHost, hs));
  HIP_TRY(ctx, hipStreamSynchronize(hs));
  const unsigned id_items = s->h_totals[0], id_bytes = s->h_totals[1], ref_bytes = s->h_totals[2];
  hipLaunchKernelGGL(k_vcf_fill, dim3(nb), dim3(TPB), 0, hs, d_text, n, L, s->off[0], s->off[1], s->off[2], s->item_off, s->values[0], s->values[2], id_items, id_bytes);
  if (id_items == 0) HIP_TRY(ctx, hipMemsetAsync(s->item_off, 0, 4, hs));
  HIP_TRY(ctx, hipGetLastError());
  out->id_list_offsets = s->off[0];
  out->id_valid = reinterpret_cast<const uint8_t*>(s->valid[0]);
  out->id_item_offsets = s->item_off;
  out->id_values = s->values[0];
  out->n_id_items = id_items;
  out->n_id_bytes = id_bytes;
  out->ref_offsets = s->off[2];
  out->ref_values = s->values[2];
  out->n_ref_bytes = ref_bytes;
  out->alt_valid = reinterpret_cast<const uint8_t*>(s->valid[1]);
  return EXON_HIP_OK;
}

int exon_text_bam(exon_hip_ctx* ctx, void* stream, ExonTextScratch** sp, const uint8_t* d_data, int64_t n_bytes, const uint32_t* d_rec_of_row, int64_t n_rows, uint64_t projection,
                  ExonBamText* out) {
  memset(out, 0, sizeof *out);
  const uint64_t all = EXON_HIP_PROJECT_BAM_NAME | EXON_HIP_PROJECT_BAM_CIGAR | EXON_HIP_PROJECT_BAM_SEQUENCE | EXON_HIP_PROJECT_BAM_QUALITY_SCORES;
  if (n_rows == 0 || !(projection & all)) return EXON_HIP_OK;
  int rc = scratch_for(ctx, sp, std::max<int64_t>(n_rows, 1 << 16), std::max<int64_t>(2 * n_bytes, 1 << 20), false);  // (a sequence doubles its 4-bit codes)
  if (rc) return rc;
  ExonTextScratch* s = *sp;
  hipStream_t hs = pick_stream(ctx, stream);
  const unsigned n = (unsigned)n_rows;
  const int nb = (int)((n + TPB - 1) / TPB);
  BamLens L{s->len[0], s->len[1], s->len[2]};
  hipLaunchKernelGGL(k_bam_measure, dim3(nb), dim3(TPB), 0, hs, d_data, d_rec_of_row, n, L, s->valid[0]);
  for (int k = 0; k < 3; ++k) scan_lengths(hs, s, s->len[k], n, s->off[k], k);
  HIP_TRY(ctx, hipMemcpyAsync(s->h_totals, s->totals, 16, hipMemcpyDeviceToHost, hs));
  HIP_TRY(ctx, hipStreamSynchronize(hs));
  const unsigned name_bytes = s->h_totals[0], cigar_bytes = s->h_totals[1], seq_bytes = s->h_totals[2];
  if ((projection & EXON_HIP_PROJECT_BAM_QUALITY_SCORES) && s->qual_cap < (size_t)seq_bytes) {
    if (s->qual) exon_pool_free(ctx, s->qual);
    s->qual_cap = std::max<size_t>((size_t)seq_bytes, (size_t)1 << 20);
    s->qual = static_cast<int64_t*>(exon_pool_alloc(ctx, s->qual_cap * 8));
    if (!s->qual) {
      s->qual_cap = 0;
      return fail(ctx, EXON_HIP_ENOMEM, "quality_scores of a slab (%u items)", seq_bytes);
    }
  }
  hipLaunchKernelGGL(k_bam_fill, dim3(nb), dim3(TPB), 0, hs, d_data, d_rec_of_row, n, projection, s->off[0], s->off[1], s->off[2], s->values[0], s->values[1], s->values[2], s->qual);
  HIP_TRY(ctx, hipGetLastError());
  out->name_offsets = s->off[0];
  out->name_values = s->values[0];
  out->name_valid = reinterpret_cast<const uint8_t*>(s->valid[0]);
  out->n_name_bytes = name_bytes;
  out->cigar_offsets = s->off[1];
  out->cigar_values = s->values[1];
  out->n_cigar_bytes = cigar_bytes;
  out->seq_offsets = s->off[2];
  out->seq_values = s->values[2];
  out->n_seq_bytes = seq_bytes;
  out->qual_values = s->qual;
  out->qual_offsets = s->off[2];  // (a BAM record's qualities are as many as its bases)
  out->n_qual_items = seq_bytes;
  return EXON_HIP_OK;
}

int exon_text_fastq(exon_hip_ctx* ctx, void* stream, ExonTextScratch** sp, const exon_hip_fastq_views* v, int64_t n_bytes, ExonFastqText* out) {
  memset(out, 0, sizeof *out);
  const int64_t n_reads = v->n_reads;
  if (n_reads == 0) return EXON_HIP_OK;
  int rc = scratch_for(ctx, sp, std::max<int64_t>(n_reads, 1 << 16), std::max<int64_t>(n_bytes, 1 << 20), false, 4);
  if (rc) return rc;
  ExonTextScratch* s = *sp;
  hipStream_t hs = pick_stream(ctx, stream);
  const unsigned n = (unsigned)n_reads;
  const int nb = (int)((n + TPB - 1) / TPB);
  FastqLens L{s->len[0], s->len[1], s->len[2], s->len[3]};
  hipLaunchKernelGGL(k_fastq_measure, dim3(nb), dim3(TPB), 0, hs, v->text_base, n, v->head_start, v->head_end, v->seq_start, v->seq_end, v->qual_start, v->qual_end, L,
                     s->valid[0]);
  for (int k = 0; k < 4; ++k) scan_lengths(hs, s, s->len[k], n, s->off[k], k);
  hipLaunchKernelGGL(k_fastq_fill, dim3(nb), dim3(TPB), 0, hs, v->text_base, n, v->head_start, v->seq_start, v->qual_start, s->off[0], s->off[1], s->off[2], s->off[3],
                     s->values[0], s->values[1], s->values[2], s->values[3]);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(s->h_totals, s->totals, 16, hipMemcpyDeviceToHost, hs));
  HIP_TRY(ctx, hipStreamSynchronize(hs));
  for (int k = 0; k < 4; ++k) {
    out->offsets[k] = s->off[k];
    out->values[k] = s->values[k];
    out->n_bytes[k] = s->h_totals[k];
  }
  out->desc_valid = reinterpret_cast<const uint8_t*>(s->valid[0]);
  return EXON_HIP_OK;
}

// SAM lines -> the BAM text columns (ExonBamText; quality_scores has list offsets of its own here: QUAL may be '*' next to a SEQ)
int exon_text_sam(exon_hip_ctx* ctx, void* stream, ExonTextScratch** sp, const uint8_t* d_text, int64_t n_bytes, const unsigned* d_nl, int64_t n_rows, uint64_t projection,
                  ExonBamText* out, int64_t* n_undecided) {
  memset(out, 0, sizeof *out);
  *n_undecided = 0;
  const uint64_t all = EXON_HIP_PROJECT_BAM_NAME | EXON_HIP_PROJECT_BAM_CIGAR | EXON_HIP_PROJECT_BAM_SEQUENCE | EXON_HIP_PROJECT_BAM_QUALITY_SCORES;
  if (n_rows == 0 || !(projection & all)) return EXON_HIP_OK;
  const unsigned skip = (unsigned)(reinterpret_cast<uintptr_t>(d_text) & 15);
  d_text -= skip;
  n_bytes += skip;
  int rc = scratch_for(ctx, sp, std::max<int64_t>(n_rows, 1 << 16), std::max<int64_t>(n_bytes, 1 << 20), false, 4);
  if (rc) return rc;
  ExonTextScratch* s = *sp;
  hipStream_t hs = pick_stream(ctx, stream);
  const unsigned n = (unsigned)n_rows;
  const int nb = (int)((n + TPB - 1) / TPB);
  SamLens L{s->len[0], s->len[1], s->len[2], s->len[3], s->sam_field_off};
  HIP_TRY(ctx, hipMemsetAsync(s->totals, 0, 16, hs));
  unsigned* d_und = s->sums + (s->max_rows + 64) / TPB + 2;  // a word of the block-sum buffer behind what scan_lengths uses (nb <= r / TPB + 1)
  HIP_TRY(ctx, hipMemsetAsync(d_und, 0, 4, hs));
  hipLaunchKernelGGL(k_sam_measure, dim3(nb), dim3(TPB), 0, hs, d_text, d_nl, n, skip, L, s->valid[0], d_und);
  for (int k = 0; k < 4; ++k) scan_lengths(hs, s, s->len[k], n, s->off[k], k);
  HIP_TRY(ctx, hipMemcpyAsync(s->h_totals, s->totals, 16, hipMemcpyDeviceToHost, hs));
  unsigned h_und = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&h_und, d_und, 4, hipMemcpyDeviceToHost, hs));
  HIP_TRY(ctx, hipStreamSynchronize(hs));
  *n_undecided = h_und;
  if (h_und) return EXON_HIP_OK;
  const unsigned qual_items = s->h_totals[3];
  if ((projection & EXON_HIP_PROJECT_BAM_QUALITY_SCORES) && s->qual_cap < (size_t)qual_items) {
    if (s->qual) exon_pool_free(ctx, s->qual);
    s->qual_cap = std::max<size_t>((size_t)qual_items, (size_t)1 << 20);
    s->qual = static_cast<int64_t*>(exon_pool_alloc(ctx, s->qual_cap * 8));
    if (!s->qual) {
      s->qual_cap = 0;
      return fail(ctx, EXON_HIP_ENOMEM, "quality_scores of a slab (%u items)", qual_items);
    }
  }
  hipLaunchKernelGGL(k_sam_fill, dim3(nb), dim3(TPB), 0, hs, d_text, n, L, projection, s->off[0], s->off[1], s->off[2], s->off[3], s->values[0], s->values[1], s->values[2],
                     s->qual);
  HIP_TRY(ctx, hipGetLastError());
  out->name_offsets = s->off[0];
  out->name_values = s->values[0];
  out->name_valid = reinterpret_cast<const uint8_t*>(s->valid[0]);
  out->n_name_bytes = s->h_totals[0];
  out->cigar_offsets = s->off[1];
  out->cigar_values = s->values[1];
  out->n_cigar_bytes = s->h_totals[1];
  out->seq_offsets = s->off[2];
  out->seq_values = s->values[2];
  out->n_seq_bytes = s->h_totals[2];
  out->qual_values = s->qual;
  out->qual_offsets = s->off[3];
  out->n_qual_items = qual_items;
  return EXON_HIP_OK;
}

int exon_text_bcf(exon_hip_ctx* ctx, void* stream, ExonTextScratch** sp, const uint8_t* d_data, int64_t n_bytes, const uint32_t* d_rec_of_row, int64_t n_rows, uint64_t projection,
                  ExonBcfText* out, int64_t* n_undecided) {
  memset(out, 0, sizeof *out);
  *n_undecided = 0;
  if (n_rows == 0 || !(projection & (EXON_HIP_PROJECT_VCF_ID | EXON_HIP_PROJECT_VCF_REF | EXON_HIP_PROJECT_VCF_ALT))) return EXON_HIP_OK;
  int rc = scratch_for(ctx, sp, std::max<int64_t>(n_rows, 1 << 16), std::max<int64_t>(n_bytes, 1 << 20), true, 5);
  if (rc) return rc;
  ExonTextScratch* s = *sp;
  hipStream_t hs = pick_stream(ctx, stream);
  const unsigned n = (unsigned)n_rows;
  const int nb = (int)((n + TPB - 1) / TPB);
  BcfLens L{s->len[0], s->len[1], s->len[2], s->len[3], s->len[4]};
  unsigned* d_und = s->totals5 + 7;
  HIP_TRY(ctx, hipMemsetAsync(s->totals5, 0, 32, hs));
  hipLaunchKernelGGL(k_bcf_measure, dim3(nb), dim3(TPB), 0, hs, d_data, d_rec_of_row, n, L, d_und);
  for (int k = 0; k < 5; ++k) {  // (scan_lengths' three launches with the totals in this call's own eight words)
    const int nbk = (int)((n + TPB - 1) / TPB);
    hipLaunchKernelGGL(k_block_sums, dim3(nbk), dim3(TPB), 0, hs, s->len[k], n, s->sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, hs, s->sums, nbk, s->totals5 + k);
    hipLaunchKernelGGL(k_write_offsets, dim3(nbk), dim3(TPB), 0, hs, s->len[k], n, s->sums, s->off[k]);
  }
  HIP_TRY(ctx, hipMemcpyAsync(s->h_totals5, s->totals5, 32, hipMemcpyDeviceToHost, hs));
  HIP_TRY(ctx, hipStreamSynchronize(hs));
  *n_undecided = s->h_totals5[7];
  if (*n_undecided) return EXON_HIP_OK;
  const unsigned id_items = s->h_totals5[0], id_bytes = s->h_totals5[1], ref_bytes = s->h_totals5[2], alt_items = s->h_totals5[3], alt_bytes = s->h_totals5[4];
  hipLaunchKernelGGL(k_bcf_fill, dim3(nb), dim3(TPB), 0, hs, d_data, d_rec_of_row, n, projection, s->off[0], s->off[1], s->off[2], s->off[3], s->off[4], s->item_off, s->item_off2,
                     s->values[0], s->values[1], s->values[2], id_items, id_bytes, alt_items, alt_bytes);
  if (id_items == 0) HIP_TRY(ctx, hipMemsetAsync(s->item_off, 0, 4, hs));
  if (alt_items == 0) HIP_TRY(ctx, hipMemsetAsync(s->item_off2, 0, 4, hs));
  HIP_TRY(ctx, hipGetLastError());
  out->id_list_offsets = s->off[0];
  out->id_item_offsets = s->item_off;
  out->id_values = s->values[0];
  out->n_id_items = id_items;
  out->n_id_bytes = id_bytes;
  out->ref_offsets = s->off[2];
  out->ref_values = s->values[1];
  out->n_ref_bytes = ref_bytes;
  out->alt_list_offsets = s->off[3];
  out->alt_item_offsets = s->item_off2;
  out->alt_values = s->values[2];
  out->n_alt_items = alt_items;
  out->n_alt_bytes = alt_bytes;
  return EXON_HIP_OK;
}
