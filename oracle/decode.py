"""ORACLE-side record decoders (TEST INFRASTRUCTURE ONLY; pure Python, small fixtures only).

They restate what the reference's batch readers + array builders put into Arrow columns, so the
oracle can be pinned on the reference's own fixture files:
  VCF   exon-vcf/src/array_builder/lazy_array_builder.rs:153-216 (chrom, pos, qual, filter, info)
        -- record parsing itself is noodles-vcf 0.70 (not in tree): tab-separated 8+ columns,
        '.' = missing, FILTER split on ';', INFO `k=v;k;...`
  BAM   exon-bam/src/array_builder.rs:102-218 over noodles-bam 0.72 records (BAM spec 4.2);
        alignment_end = start + sum(len of M/D/N/=/X ops) - 1 (exon-bam/src/indexed_async_batch_stream.rs:45-64)
  FASTQ exon-fastq/src/array_builder.rs:68-102    FASTA exon-fasta/src/array_builder.rs:114-132
The product has its own C++ decoders (exon_amd/csrc/host); nothing here is shipped.
"""
import gzip
import struct

import numpy as np


def read_bytes(path):
    raw = open(path, "rb").read()
    if raw[:2] == b"\x1f\x8b":  # gzip / BGZF (multi-member): python's gzip reads all members
        return gzip.decompress(raw)
    return raw


# ---- VCF ---------------------------------------------------------------------------------------------
def decode_vcf(path):
    """-> dict(contigs, filters_header, info_header, chrom[str], pos[int|None], qual[np.float32|None],
    filter[list[str]], info[dict|None])"""
    text = read_bytes(path).decode()
    contigs, filt_hdr, info_hdr = [], [], {}
    rows = dict(chrom=[], pos=[], qual=[], filter=[], info=[])
    for line in text.split("\n"):
        if not line:
            continue
        if line.startswith("##"):
            if line.startswith("##contig=<"):
                contigs.append(_hdr_fields(line)["ID"])
            elif line.startswith("##FILTER=<"):
                filt_hdr.append(_hdr_fields(line)["ID"])
            elif line.startswith("##INFO=<"):
                f = _hdr_fields(line)
                info_hdr[f["ID"]] = (f.get("Number"), f.get("Type"))
            continue
        if line.startswith("#"):
            continue
        c = line.split("\t")
        rows["chrom"].append(c[0])
        # noodles-vcf lazy record: POS "0" (telomere) has no variant_start -> None; anything else must parse as an
        # unsigned integer or `record.variant_start().transpose()?` fails (lazy_array_builder.rs:163-168)
        if not c[1].isascii() or not c[1].isdigit():
            raise ValueError(f"invalid POS {c[1]!r}")
        rows["pos"].append(int(c[1]) or None)
        rows["qual"].append(None if c[5] == "." else np.float32(c[5]))
        rows["filter"].append([] if c[6] == "." else c[6].split(";"))
        if c[7] == ".":
            rows["info"].append(None)
        else:
            d = {}
            for kv in c[7].split(";"):
                k, _, v = kv.partition("=")
                if k not in d:  # `Info::get` returns the first field whose key matches (a repeated key is malformed VCF anyway)
                    d[k] = v if _ else True
            rows["info"].append(d)
    rows.update(contigs=contigs, filters_header=filt_hdr, info_header=info_hdr)
    return rows


def _hdr_fields(line):
    body = line[line.index("<") + 1:line.rindex(">")]
    out, cur, inq, key = {}, "", False, None
    for ch in body + ",":
        if ch == '"':
            inq = not inq
        elif ch == "=" and not inq and key is None:
            key, cur = cur, ""
        elif ch == "," and not inq:
            out[key] = cur
            key, cur = None, ""
        else:
            cur += ch
    return out


def vcf_device_columns(v, contigs=None):
    """chrom dictionary ids in header-contig order (unknown names appended in order of appearance)."""
    names = list(contigs if contigs is not None else v["contigs"])
    for c in v["chrom"]:
        if c not in names:
            names.append(c)
    idx = {n: i for i, n in enumerate(names)}
    chrom_id = np.array([idx[c] for c in v["chrom"]], np.int32)
    pos = np.array([p if p is not None else 0 for p in v["pos"]], np.int64)
    pos_valid = np.packbits(np.array([p is not None for p in v["pos"]], bool), bitorder="little")
    return names, chrom_id, pos, pos_valid


# ---- BAM ---------------------------------------------------------------------------------------------
CIGAR_OPS = "MIDNSHP=X"
REF_CONSUMING = {0, 2, 3, 7, 8}


def decode_bam(path):
    b = read_bytes(path)
    assert b[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", b, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", b, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", b, o)
        name = b[o + 4:o + 4 + l_name - 1].decode()
        l_ref, = struct.unpack_from("<i", b, o + 4 + l_name)
        refs.append((name, l_ref))
        o += 8 + l_name
    recs = []
    while o < len(b):
        block, = struct.unpack_from("<i", b, o)
        (ref_id, pos, l_read_name, mapq, _bin, n_cigar, flag, l_seq, next_ref, next_pos, tlen) = struct.unpack_from(
            "<iiBBHHHiiii", b, o + 4)
        p = o + 36
        name = b[p:p + l_read_name - 1].decode()
        p += l_read_name
        cigar = struct.unpack_from(f"<{n_cigar}I", b, p)
        p += 4 * n_cigar
        seq = b[p:p + (l_seq + 1) // 2]
        p += (l_seq + 1) // 2
        qual = list(b[p:p + l_seq])
        ref_len = sum(c >> 4 for c in cigar if (c & 0xF) in REF_CONSUMING)
        start = pos + 1 if pos >= 0 else None
        recs.append(dict(
            name=name, flag=flag, ref_id=ref_id if ref_id >= 0 else None, start=start,
            end=(start + ref_len - 1) if start is not None else None,
            mapq=None if mapq == 255 else mapq,
            cigar="".join(f"{c >> 4}{CIGAR_OPS[c & 0xF]}" for c in cigar),
            mate_ref_id=next_ref if next_ref >= 0 else None,
            sequence="".join("=ACMGRSVTWYHKDBN"[(seq[i >> 1] >> (4 if i % 2 == 0 else 0)) & 0xF] for i in range(l_seq)),
            quality_score=qual))
        o += 4 + block
    return refs, recs


def bam_device_columns(recs):
    n = len(recs)
    flag = np.array([r["flag"] for r in recs], np.int32)
    mapq = np.array([r["mapq"] if r["mapq"] is not None else 255 for r in recs], np.uint8)
    mv = np.packbits(np.array([r["mapq"] is not None for r in recs], bool), bitorder="little")
    ref = np.array([r["ref_id"] if r["ref_id"] is not None else -1 for r in recs], np.int32)
    rv = np.packbits(np.array([r["ref_id"] is not None for r in recs], bool), bitorder="little")
    return n, flag, mapq, mv, ref, rv


# ---- SAM (text) --------------------------------------------------------------------------------------
def decode_sam(path):
    """Same columns as BAM (exon-sam/src/schema_builder.rs:371-402): RNAME through the @SQ order ('*' -> None), POS 0 ->
    None, MAPQ 255 -> None, end from the CIGAR."""
    refs, recs = [], []
    for line in read_bytes(path).decode().split("\n"):
        if not line:
            continue
        if line.startswith("@"):
            if line.startswith("@SQ"):
                f = dict(x.split(":", 1) for x in line.split("\t")[1:] if ":" in x)
                refs.append((f["SN"], int(f.get("LN", 0))))
            continue
        c = line.split("\t")
        names = [r[0] for r in refs]
        pos, mapq = int(c[3]), int(c[4])
        ref_len, num = 0, ""
        if c[5] != "*":
            for ch in c[5]:
                if ch.isdigit():
                    num += ch
                else:
                    if CIGAR_OPS.index(ch) in REF_CONSUMING:
                        ref_len += int(num)
                    num = ""
        start = pos if pos >= 1 else None
        recs.append(dict(name=c[0], flag=int(c[1]), ref_id=names.index(c[2]) if c[2] in names else None, start=start,
                         end=(start + ref_len - 1) if start is not None else None, mapq=None if mapq == 255 else mapq,
                         cigar=c[5]))
    return refs, recs


# ---- BCF2 ----------------------------------------------------------------------------------------------
def decode_bcf(path):
    """BCF 2.2 (VCF specification section 6) -> the same dict as decode_vcf: CHROM through the header contigs, POS = pos0 + 1
    (pos0 = -1 -> None), QUAL 0x7F800001 -> None, FILTER / INFO keys through the header string dictionary (PASS = 0, then
    FILTER / INFO / FORMAT IDs in order of first appearance unless IDX= says otherwise)."""
    b = read_bytes(path)
    assert b[:5] == b"BCF\x02\x02"
    l_text, = struct.unpack_from("<I", b, 5)
    text = b[9:9 + l_text].rstrip(b"\0").decode()
    contigs, strings, filt_hdr, info_hdr = [], {"PASS": 0}, [], {}
    order = ["PASS"]
    for line in text.split("\n"):
        if line.startswith("##contig=<"):
            contigs.append(_hdr_fields(line)["ID"])
        elif line.startswith(("##FILTER=<", "##INFO=<", "##FORMAT=<")):
            f = _hdr_fields(line)
            if line.startswith("##FILTER=<"):
                filt_hdr.append(f["ID"])
            if line.startswith("##INFO=<"):
                info_hdr[f["ID"]] = (f.get("Number"), f.get("Type"))
            if f["ID"] not in strings:
                strings[f["ID"]] = int(f["IDX"]) if "IDX" in f else len(order)
                order.append(f["ID"])
    by_idx = {v: k for k, v in strings.items()}
    rows = dict(chrom=[], pos=[], qual=[], filter=[], info=[])
    size = {1: 1, 2: 2, 3: 4, 5: 4, 7: 1}
    fmt = {1: "b", 2: "h", 3: "i", 5: "f"}
    missing = {1: -128, 2: -32768, 3: -2147483648}

    def typed(o):
        d = b[o]
        o += 1
        n, t = d >> 4, d & 15
        if n == 15:
            (cnt, o2) = typed(o)
            n, o = cnt[0], o2
        if t == 0:
            return [], o
        if t == 7:
            return b[o:o + n], o + n
        vals = list(struct.unpack_from(f"<{n}{fmt[t]}", b, o))
        if t == 5:
            raw = struct.unpack_from(f"<{n}I", b, o)
            vals = [None if r == 0x7F800001 else v for v, r in zip(vals, raw) if r != 0x7F800002]
        else:
            vals = [None if v == missing[t] else v for v in vals if v != missing[t] + 1]
        return vals, o + n * size[t]

    o = 9 + l_text
    while o < len(b):
        l_shared, l_indiv = struct.unpack_from("<II", b, o)
        r = o + 8
        chrom, pos0, _rlen, qbits, nia, _nfs = struct.unpack_from("<iiiIII", b, r)
        n_info, n_allele = nia & 0xFFFF, nia >> 16
        p = r + 24
        _id, p = typed(p)
        for _ in range(n_allele):
            _a, p = typed(p)
        filt, p = typed(p)
        info = {}
        for _ in range(n_info):
            key, p = typed(p)
            val, p = typed(p)
            name = by_idx[key[0]]
            if isinstance(val, (bytes, bytearray)):
                info[name] = val.decode()
            elif len(val) == 0:
                info[name] = True
            else:
                info[name] = val[0] if len(val) == 1 else val
        rows["chrom"].append(contigs[chrom])
        rows["pos"].append(pos0 + 1 if pos0 >= 0 else None)
        rows["qual"].append(None if qbits == 0x7F800001 else np.frombuffer(struct.pack("<I", qbits), np.float32)[0])
        rows["filter"].append([by_idx[i] for i in filt])
        rows["info"].append(info if n_info else None)
        o += 8 + l_shared + l_indiv
    rows.update(contigs=contigs, filters_header=filt_hdr, info_header=info_hdr)
    return rows


# ---- FASTQ / FASTA -----------------------------------------------------------------------------------
def decode_fastq(path):
    lines = read_bytes(path).decode().split("\n")
    recs = []
    i = 0
    while i + 3 < len(lines) and lines[i].startswith("@"):
        head = lines[i][1:]
        name, _, desc = head.partition(" ")
        recs.append(dict(name=name, description=desc if desc else None, sequence=lines[i + 1],
                         quality_scores=lines[i + 3]))
        i += 4
    return recs


def fastq_device_columns(recs):
    q = [r["quality_scores"].encode() for r in recs]
    off = np.zeros(len(q) + 1, np.int32)
    off[1:] = np.cumsum([len(x) for x in q])
    return off, np.frombuffer(b"".join(q), np.uint8).copy()


def decode_fasta(path):
    recs = []
    for line in read_bytes(path).decode().split("\n"):
        if line.startswith(">"):
            head = line[1:]
            parts = head.split(None, 1)
            recs.append(dict(id=parts[0] if parts else "", description=parts[1] if len(parts) > 1 else None,
                             sequence=""))
        elif line and recs:
            recs[-1]["sequence"] += line.strip()
    return recs
