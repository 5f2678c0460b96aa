"""Test infrastructure: tabix (.tbi) and BAI (.bai) index writers for BGZF files made by the tests.

The image has no htslib tools, and the reference's fixtures with indexes are a few hundred records (one BGZF block), so
the chunk arithmetic of an indexed scan -- start inside a block, end inside another, many blocks in between -- needs
files of our own.  The formats are the published ones (SAM specification 5.2 "The BAI index format", tabix: Li 2011,
"TBI format"): UCSC binning (min_shift 14, depth 5), per bin a list of chunks of BGZF virtual offsets, per reference a
linear index of 16 KiB windows.  Not product code and not the oracle: only tests import it.
"""
import bisect
import gzip
import struct
import zlib

import numpy as np


def bgzf_blocks(raw):
    """[(compressed offset, block size, inflated bytes)] of every block of a BGZF file."""
    out, o = [], 0
    while o < len(raw):
        assert raw[o:o + 4] == b"\x1f\x8b\x08\x04", "not a BGZF block"
        xlen = struct.unpack_from("<H", raw, o + 10)[0]
        x, bsize = o + 12, None
        while x < o + 12 + xlen:
            si1, si2, slen = raw[x], raw[x + 1], struct.unpack_from("<H", raw, x + 2)[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", raw, x + 4)[0] + 1
            x += 4 + slen
        data = zlib.decompress(raw[o + 12 + xlen:o + bsize - 8], -15)
        out.append((o, bsize, data))
        o += bsize
    return out


class VirtualOffsets:
    """global uncompressed offset -> BGZF virtual offset"""

    def __init__(self, blocks):
        self.ustart, self.coff, u = [], [], 0
        for coff, _bsize, data in blocks:
            if not data:
                continue
            self.ustart.append(u)
            self.coff.append(coff)
            u += len(data)
        self.total = u
        self.end_coff = blocks[-1][0] + blocks[-1][1] if blocks else 0

    def at(self, u):
        if u >= self.total:
            return self.end_coff << 16
        i = bisect.bisect_right(self.ustart, u) - 1
        return (self.coff[i] << 16) | (u - self.ustart[i])


def reg2bin(beg, end):
    end -= 1
    if beg >> 14 == end >> 14:
        return 4681 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return 585 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return 73 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return 9 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return 1 + (beg >> 26)
    return 0


class RefIndex:
    def __init__(self):
        self.bins, self.linear = {}, []

    def add(self, beg, end, v0, v1):
        chunks = self.bins.setdefault(reg2bin(beg, end), [])
        if chunks and chunks[-1][1] == v0:
            chunks[-1][1] = v1
        else:
            chunks.append([v0, v1])
        w1 = (end - 1) >> 14
        while len(self.linear) <= w1:
            self.linear.append(0)
        for w in range(beg >> 14, w1 + 1):
            if self.linear[w] == 0 or v0 < self.linear[w]:
                self.linear[w] = v0

    def pack(self):
        for i in range(1, len(self.linear)):  # windows without a record take their predecessor's offset (htslib)
            if self.linear[i] == 0:
                self.linear[i] = self.linear[i - 1]
        out = struct.pack("<i", len(self.bins))
        for b in sorted(self.bins):
            out += struct.pack("<Ii", b, len(self.bins[b]))
            for v0, v1 in self.bins[b]:
                out += struct.pack("<QQ", v0, v1)
        out += struct.pack("<i", len(self.linear)) + b"".join(struct.pack("<Q", v) for v in self.linear)
        return out


def write_tabix(vcf_gz):
    """<vcf_gz>.tbi for a BGZF-compressed, coordinate-sorted VCF; returns the number of records indexed."""
    raw = open(vcf_gz, "rb").read()
    blocks = bgzf_blocks(raw)
    vo = VirtualOffsets(blocks)
    text = b"".join(d for _, _, d in blocks)
    names, refs, n, u = [], {}, 0, 0
    while u < len(text):
        e = text.find(b"\n", u)
        e = len(text) if e < 0 else e + 1
        line = text[u:e]
        if line[:1] != b"#" and line.strip():
            f = line.split(b"\t", 5)
            name, pos, ref = f[0].decode(), int(f[1]), f[3]
            if name not in refs:
                names.append(name)
                refs[name] = RefIndex()
            refs[name].add(pos - 1, pos - 1 + max(1, len(ref)), vo.at(u), vo.at(e))
            n += 1
        u = e
    nm = b"".join(x.encode() + b"\0" for x in names)
    body = b"TBI\1" + struct.pack("<8i", len(names), 2, 1, 2, 0, ord("#"), 0, len(nm)) + nm
    body += b"".join(refs[x].pack() for x in names)
    with gzip.open(str(vcf_gz) + ".tbi", "wb") as fh:
        fh.write(body)
    return n


def write_bai(bam):
    """<bam>.bai for a coordinate-sorted BAM; returns the number of records indexed."""
    raw = open(bam, "rb").read()
    blocks = bgzf_blocks(raw)
    vo = VirtualOffsets(blocks)
    data = b"".join(d for _, _, d in blocks)
    assert data[:4] == b"BAM\1"
    l_text = struct.unpack_from("<i", data, 4)[0]
    o = 8 + l_text
    n_ref = struct.unpack_from("<i", data, o)[0]
    o += 4
    for _ in range(n_ref):
        l_name = struct.unpack_from("<i", data, o)[0]
        o += 4 + l_name + 4
    refs = [RefIndex() for _ in range(n_ref)]
    n = 0
    while o < len(data):
        bs = struct.unpack_from("<i", data, o)[0]
        ref_id, pos, l_name, _mapq, _bin, n_cig = struct.unpack_from("<iiBBHH", data, o + 4)
        span = 0
        for k in range(n_cig):
            c = struct.unpack_from("<I", data, o + 36 + l_name + 4 * k)[0]
            if (c & 15) in (0, 2, 3, 7, 8):
                span += c >> 4
        if ref_id >= 0 and pos >= 0:
            refs[ref_id].add(pos, pos + max(1, span), vo.at(o), vo.at(o + 4 + bs))
            n += 1
        o += 4 + bs
    with open(str(bam) + ".bai", "wb") as fh:
        fh.write(b"BAI\1" + struct.pack("<i", n_ref) + b"".join(r.pack() for r in refs))
    return n


def sorted_bam(path_ubam, n, rng, n_ref=4):
    """A coordinate-sorted uncompressed BAM: n records over n_ref references plus unmapped ones at the end."""
    refs = [(f"chr{i + 1}".encode(), 50_000_000) for i in range(n_ref)]
    text = b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % r for r in refs)
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", n_ref)
    for name, ln in refs:
        head += struct.pack("<i", len(name) + 1) + name + b"\0" + struct.pack("<i", ln)
    ref = np.sort(rng.integers(0, n_ref, n))
    pos = rng.integers(0, 49_000_000, n)
    order = np.lexsort((pos, ref))
    ref, pos = ref[order], pos[order]
    n_unmapped = n // 50
    out = [head]
    rows = []
    for i in range(n):
        unm = i >= n - n_unmapped
        r, p = (-1, -1) if unm else (int(ref[i]), int(pos[i]))
        ln = int(rng.integers(30, 151))
        name = b"r%d\0" % i
        flag = 4 if unm else int(rng.choice([99, 147, 83, 163, 1123]))
        mapq = int(rng.choice([0, 20, 40, 60, 255]))
        body = struct.pack("<iiBBHHHiiii", r, p, len(name), mapq, 4680, 1, flag, ln, -1, -1, 0)
        body += name + struct.pack("<I", (ln << 4) | 0) + bytes((ln + 1) // 2) + bytes(ln)
        out.append(struct.pack("<i", len(body)) + body)
        rows.append((r, p + 1, p + ln))
    open(path_ubam, "wb").write(b"".join(out))
    return rows
