"""ORACLE-side record decoders (TEST INFRASTRUCTURE ONLY; pure Python, small fixtures only).

They restate what the reference's batch readers + array builders put into Arrow columns, so the
oracle can be pinned on the reference's own fixture files:
  VCF   exon-vcf/src/array_builder/lazy_array_builder.rs:153-216 (chrom, pos, qual, filter, info)
        -- record parsing itself is noodles-vcf 0.70 (not in tree): tab-separated 8+ columns,
        '.' = missing, FILTER split on ';', INFO `k=v;k;...`
  BAM   exon-bam/src/array_builder.rs:102-218 over noodles-bam 0.72 records (BAM spec 4.2);
        alignment_end = start + sum(len of M/D/N/=/X ops) - 1 (exon-bam/src/indexed_async_batch_stream.rs:45-64)
  FASTQ exon-fastq/src/array_builder.rs:68-102    FASTA exon-fasta/src/array_builder.rs:114-132
  CRAM  exon-cram/src/{async_batch_stream,array_builder}.rs over noodles-cram (not in tree): CRAM 3.0 restated from the
        specification and pinned on the reference's slt first rows + container record counts (tests/test_cram.py); the
        CRAM 3.1 rANS Nx16 block codec (`_rans_nx16`) is PARITY UNPINNED -- no 3.1 stream exists in the reference's
        fixtures or in this image; see the note above it and DESIGN.md section 7j
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
    filter[list[str]], info[dict|None], id[list[str]|None], ref[str], alt[list|None])
    id / ref / alt follow LazyVCFArrayBuilder::append (exon-vcf/src/array_builder/lazy_array_builder.rs:169-205): ids NULL when the
    field is '.', else its ';'-separated items; ref the field; alt NULL when '.', else a list WITHOUT items -- the reference
    builds a string of the alternate bases and then calls `alternates.append(true)` without appending it (:191-205)."""
    text = read_bytes(path).decode()
    contigs, filt_hdr, info_hdr, format_hdr = [], [], {}, {}
    rows = dict(chrom=[], pos=[], qual=[], filter=[], info=[], id=[], ref=[], alt=[], info_raw=[], format_raw=[], samples_raw=[])
    for line in text.split("\n"):
        line = line[:-1] if line.endswith("\r") else line  # the line reader drops a CR in front of the LF (noodles' read_line)
        if not line:
            continue
        if line.startswith("##"):
            if line.startswith("##contig=<"):
                contigs.append(_hdr_fields(line)["ID"])
            elif line.startswith("##FILTER=<"):
                filt_hdr.append(_hdr_fields(line)["ID"])
            elif line.startswith("##INFO=<"):
                f = _hdr_fields(line)
                info_hdr.setdefault(f["ID"], (f.get("Number"), f.get("Type")))
            elif line.startswith("##FORMAT=<"):
                f = _hdr_fields(line)
                format_hdr.setdefault(f["ID"], (f.get("Number"), f.get("Type")))
            continue
        if line.startswith("#"):
            continue
        c = line.split("\t")
        rows["chrom"].append(c[0])
        # noodles-vcf lazy record: POS "0" (telomere) has no variant_start -> None; anything else must parse as an
        # unsigned integer or `record.variant_start().transpose()?` fails (lazy_array_builder.rs:163-168)
        # (usize::from_str: one optional '+', then digits -- noodles-vcf 0.70 has no number parser among its dependencies,
        # Cargo.lock:3915-3930, so POS goes through core's FromStr)
        digits = c[1][1:] if c[1].startswith("+") else c[1]
        if not digits.isascii() or not digits.isdigit():
            raise ValueError(f"invalid POS {c[1]!r}")
        rows["pos"].append(int(digits) or None)
        rows["qual"].append(None if c[5] == "." else np.float32(c[5]))
        rows["filter"].append([] if c[6] == "." else c[6].split(";"))
        rows["id"].append(None if c[2] in (".", "") else c[2].split(";"))
        rows["ref"].append(c[3])
        rows["alt"].append(None if c[4] in (".", "") else [])
        rows["info_raw"].append(c[7])
        rows["format_raw"].append(c[8] if len(c) > 8 else None)
        rows["samples_raw"].append(c[9:])
        if c[7] == ".":
            rows["info"].append(None)
        else:
            d = {}
            for kv in c[7].split(";"):
                k, _, v = kv.partition("=")
                if k not in d:  # `Info::get` returns the first field whose key matches (a repeated key is malformed VCF anyway)
                    d[k] = v if _ else True
            rows["info"].append(d)
    rows.update(contigs=contigs, filters_header=filt_hdr, info_header=info_hdr, format_header=format_hdr)
    return rows


# reserved keys of the VCF specification (4.3, tables 1 and 2 + the structural-variant keys): the types noodles takes for a key
# the header does not declare (then String)
_RESERVED_INFO = dict.fromkeys("AC AD ADF ADR AN DP END MQ0 NS SB SVLEN CIPOS CIEND HOMLEN CILEN DPADJ CN CNADJ CICN CICNADJ".split(), "Integer")
_RESERVED_INFO.update(dict.fromkeys("AF BQ MQ".split(), "Float"))
_RESERVED_INFO.update(dict.fromkeys("DB H2 H3 SOMATIC VALIDATED 1000G IMPRECISE NOVEL".split(), "Flag"))
_RESERVED_FORMAT = dict.fromkeys("AD ADF ADR DP EC GQ HQ MQ PL PP PQ PS CN NQ HAP AHAP".split(), "Integer")
_RESERVED_FORMAT.update(dict.fromkeys("GL GP CNQ CNL CNP".split(), "Float"))


def rust_f32_display(x):
    """Rust's `{}` of an f32 (`v.to_string()`, lazy_array_builder.rs:233): shortest digits that round-trip, never an exponent."""
    v = np.float32(x)
    if np.isnan(v):
        return "NaN"
    if np.isinf(v):
        return "-inf" if v < 0 else "inf"
    return np.format_float_positional(v, unique=True, trim="-")


def _print_value(typ, text):
    if typ in ("String", None) or (typ == "Character" and "," not in text):
        return text
    out = []
    for item in text.split(","):
        if item == ".":
            if typ != "Character":  # a Character list drops its missing items (lazy_array_builder.rs:364-368 / :238-250 keeps them for INFO ...)
                out.append(".")
        elif typ == "Integer":
            if not (item.isascii() and item.lstrip("+-").isdigit() and len(item) - len(item.lstrip("+-")) <= 1):
                raise ValueError(f"invalid integer {item!r}")
            if not -2**31 <= int(item) < 2**31:
                raise ValueError(f"integer out of the int32 range {item!r}")
            out.append(str(int(item)))
        elif typ == "Float":
            out.append(rust_f32_display(np.float32(item)))
        else:
            out.append(item)
    return ",".join(out)


def info_string(v, row):
    """The `info` Utf8 column (parse_info = false) of row `row` of a `decode_vcf` result: the parsed entries printed again
    (exon-vcf/src/array_builder/lazy_array_builder.rs:216-297): key=value joined by ';', a Flag as key=true, Integer / Float items
    through Rust's Display, missing items '.', INFO '.' -> the empty string; a missing value is the reference's unwrap panic."""
    raw = v["info_raw"][row]
    if raw in (".", ""):
        return ""
    out = []
    for kv in raw.split(";"):
        if not kv:
            continue
        k, eq, val = kv.partition("=")
        typ = v["info_header"].get(k, (None, _RESERVED_INFO.get(k, "String")))[1]
        if typ == "Flag":
            out.append(k + "=true")
            continue
        if not eq or val in ("", "."):
            raise ValueError(f"INFO key {k!r} has no value (the reference panics)")
        out.append(k + "=" + _print_value(typ, val))
    return ";".join(out)


def formats_string(v, row):
    """The `formats` Utf8 column (parse_formats = false): FORMAT keys joined by ':', TAB, samples joined by TAB, each sample's
    values printed like INFO values and joined by ':' (lazy_array_builder.rs:310-423).  A genotype prints allele 0, then for
    every further allele the separator of the allele BEFORE it (:331-360); allele 0's own phasing is, before VCF 4.4, unphased
    when any separator of the genotype is '/' (noodles-vcf 0.70 as published; not in /root/reference)."""
    fmt = v["format_raw"][row]
    keys = [] if fmt in (None, ".", "") else fmt.split(":")
    head = ":".join(keys)
    if not keys:
        return head + "\t"
    samples = []
    for stext in v["samples_raw"][row]:
        vals = []
        for k, val in zip(keys, stext.split(":")):
            if val in ("", "."):
                raise ValueError("a sample value is missing (the reference panics)")
            if k == "GT":
                alleles, seps, cur = [], [], ""
                for ch in val:
                    if ch in "/|":
                        alleles.append(cur)
                        seps.append(ch)
                        cur = ""
                    else:
                        cur += ch
                alleles.append(cur)
                phasing = ["/" if "/" in val else "|"] + seps  # phasing[i] belongs to allele i
                txt = ""
                for i, a in enumerate(alleles):
                    a = a if a == "." else str(int(a))
                    txt += a if i == 0 else phasing[i - 1] + a
                vals.append(txt)
            else:
                typ = v["format_header"].get(k, (None, _RESERVED_FORMAT.get(k, "String")))[1]
                vals.append(_print_value(typ, val))
        samples.append(":".join(vals))
    return head + "\t" + "\t".join(samples)


def typed_info(v, key):
    """info.<key> of a decoded VCF / BCF (`decode_vcf` / `decode_bcf`) the way the reference types and builds it:
    schema (exon-core/src/datasources/vcf/schema_builder.rs:197-249): Integer -> Int32, Float -> Float32, Flag -> Boolean,
    String / Character -> Utf8; Number = 0 | 1 -> the scalar, any other Number -> List<item>.
    values (exon-vcf/src/array_builder/info_builder.rs:152-309): INFO '.' (NULL struct), key absent or value '.' -> NULL;
    a Flag is true by being there; list elements '.' -> NULL items; an unparsable value is the record's parse error.
    -> (arrow type as text, python list of None | int | float (f32-rounded) | True | str | list)"""
    number, typ = v["info_header"][key]
    scalar = number in ("0", "1")
    item = {"Integer": "int32", "Float": "float", "Flag": "bool", "String": "string", "Character": "string"}[typ]

    def conv(x):
        if x is None or x == ".":
            return None
        if typ == "Integer":
            if isinstance(x, str):
                if not (x.lstrip("+-").isdigit() and x.isascii() and len(x) - len(x.lstrip("+-")) <= 1):
                    raise ValueError(f"invalid INFO integer {x!r}")
                x = int(x)
            if isinstance(x, float) or not -2**31 <= x < 2**31:
                raise ValueError(f"INFO integer out of the int32 range: {x!r}")
            return int(x)
        if typ == "Float":
            return float(np.float32(x))
        return str(x)

    col = []
    for i in v["info"]:
        x = None if i is None else i.get(key)
        if typ == "Flag":
            col.append(True if x is not None else None)
        elif x is None or x is True or x == "." or x == "":
            col.append(None)
        elif scalar:
            col.append(conv(x))
        else:
            items = x.split(",") if isinstance(x, str) else (list(x) if isinstance(x, (list, tuple)) else [x])
            col.append([conv(e) for e in items])
    return (item if scalar else f"list<item: {item}>"), col


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
                         cigar="" if c[5] == "*" else c[5],
                         # exon-sam/src/array_builder.rs:101-185 over noodles' RecordBuf: QNAME '*' -> None, SEQ '*' -> "",
                         # QUAL '*' -> [], else Phred = char - 33
                         name_opt=None if c[0] == "*" else c[0], sequence="" if c[9] == "*" else c[9],
                         quality_score=[] if c[10] == "*" else [ord(ch) - 33 for ch in c[10]]))
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
    # id / ref / alt: BCF records go through the reference's EAGER builder (exon-bcf/src/batch_reader.rs:72 ->
    # exon-vcf/src/array_builder/eager_array_builder.rs:112-134): lists with their items, never NULL
    rows = dict(chrom=[], pos=[], qual=[], filter=[], info=[], id=[], ref=[], alt=[])
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
        alleles = []
        for _ in range(n_allele):
            _a, p = typed(p)
            alleles.append(bytes(_a).decode())
        ids = bytes(_id).decode()
        rows["id"].append([] if ids in ("", ".") else ids.split(";"))
        rows["ref"].append(alleles[0] if alleles else "")
        rows["alt"].append(alleles[1:])
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



def decode_fastq(path):
    # (the line reader drops a CR in front of the LF, as for VCF)
    lines = [ln[:-1] if ln.endswith("\r") else ln for ln in read_bytes(path).decode().split("\n")]
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


# ---- CRAM 3.0 ------------------------------------------------------------------------------------------
# Reference: exon-cram/src/array_builder.rs (columns of the shared SAM/BAM/CRAM schema, exon-sam/src/schema_builder.rs:371-402)
# over noodles-cram 0.x (Cargo.lock), which is absent from /root/reference: restated from the published CRAM 3.0
# specification (file definition, containers, compression header, slices, blocks, data-series encodings, record layout).
# Block codecs: raw, gzip, bzip2, lzma (python's own modules) and rANS 4x8 (what the reference's fixtures use); the CRAM 3.1 codecs are reported as unsupported.
def _itf8(b, o):
    v = b[o]
    if v < 0x80:
        return v, o + 1
    if v < 0xC0:
        return ((v & 0x3F) << 8) | b[o + 1], o + 2
    if v < 0xE0:
        return ((v & 0x1F) << 16) | (b[o + 1] << 8) | b[o + 2], o + 3
    if v < 0xF0:
        return ((v & 0x0F) << 24) | (b[o + 1] << 16) | (b[o + 2] << 8) | b[o + 3], o + 4
    return ((v & 0x0F) << 28) | (b[o + 1] << 20) | (b[o + 2] << 12) | (b[o + 3] << 4) | (b[o + 4] & 0x0F), o + 5


def _itf8s(b, o):
    v, o = _itf8(b, o)
    return (v - (1 << 32) if v >= (1 << 31) else v), o


def _ltf8(b, o):
    v = b[o]
    n = 0
    while n < 8 and (v << n) & 0x80:
        n += 1
    val = v & (0xFF >> (n + 1)) if n < 7 else 0
    for i in range(n):
        val = (val << 8) | b[o + 1 + i]
    return val, o + 1 + n


def _rans_freqs(b, o):
    """one frequency table of rANS 4x8: symbols run-length coded, frequencies 1 or 2 bytes, total 4096"""
    F = [0] * 256
    sym, rle = b[o], 0
    o += 1
    last = sym
    while True:
        f = b[o]
        o += 1
        if f >= 128:
            f = ((f & 127) << 8) | b[o]
            o += 1
        F[sym] = f
        if rle:
            rle -= 1
            sym += 1
        else:
            sym = b[o]
            o += 1
            if sym == last + 1:
                rle = b[o]
                o += 1
        last = sym
        if sym == 0:
            break
    C = [0] * 257
    for i in range(256):
        C[i + 1] = C[i] + F[i]
    lut = []
    for i in range(256):
        lut += [i] * F[i]
    lut += [0] * (4096 - len(lut))
    return F, C, lut, o


def _rans_4x8(b):
    """CRAM 3.0 rANS codec (4 interleaved states, 12-bit frequencies, byte renormalisation), orders 0 and 1"""
    order = b[0]
    _csz, n = struct.unpack_from("<II", b, 1)
    o = 9
    out = bytearray(n)
    if order == 0:
        F, C, lut, o = _rans_freqs(b, o)
        R = list(struct.unpack_from("<4I", b, o))
        o += 16
        for i in range(n):
            j = i & 3
            f = R[j] & 0xFFF
            sy = lut[f]
            out[i] = sy
            R[j] = F[sy] * (R[j] >> 12) + f - C[sy]
            while R[j] < (1 << 23):
                R[j] = (R[j] << 8) | b[o]
                o += 1
        return bytes(out)
    tabs = {}
    ctx, rle = b[o], 0
    o += 1
    last = ctx
    while True:
        F, C, lut, o = _rans_freqs(b, o)
        tabs[ctx] = (F, C, lut)
        if rle:
            rle -= 1
            ctx += 1
        else:
            ctx = b[o]
            o += 1
            if ctx == last + 1:
                rle = b[o]
                o += 1
        last = ctx
        if ctx == 0:
            break
    R = list(struct.unpack_from("<4I", b, o))
    o += 16
    q = n >> 2
    idx = [0, q, 2 * q, 3 * q]
    prev = [0, 0, 0, 0]
    for k in range(q):
        for j in range(4):
            F, C, lut = tabs[prev[j]]
            f = R[j] & 0xFFF
            sy = lut[f]
            out[idx[j] + k] = sy
            R[j] = F[sy] * (R[j] >> 12) + f - C[sy]
            while R[j] < (1 << 23):
                R[j] = (R[j] << 8) | b[o]
                o += 1
            prev[j] = sy
    for i in range(4 * q, n):  # the remainder belongs to the last state
        F, C, lut = tabs[prev[3]]
        f = R[3] & 0xFFF
        sy = lut[f]
        out[i] = sy
        R[3] = F[sy] * (R[3] >> 12) + f - C[sy]
        while R[3] < (1 << 23):
            R[3] = (R[3] << 8) | b[o]
            o += 1
        prev[3] = sy
    return bytes(out)


# ---- rANS Nx16 (CRAM 3.1 block method 5; "CRAM codecs" specification, section 3) ------------------------------------------
# Restated from the published format (noodles-cram, through which exon-cram/src/async_batch_stream.rs reads CRAM, is not in
# the tree): flags byte -- 0x01 order-1, 0x04 32 states instead of 4, 0x08 striped, 0x10 size not stored, 0x20 stored as is,
# 0x40 run-length coded, 0x80 bit-packed -- then sizes as big-endian base-128 integers.  PARITY UNPINNED against htslib: no CRAM 3.1
# file exists here; the decoder is pinned on streams of the test-side encoder (tests/cram_writer.py), and both it and the
# product insist that every rANS state ends at the encoder's initial value 2^15, which a misread format does not survive.
class _Bytes:
    def __init__(self, b, o=0):
        self.b, self.o = b, o

    def u8(self):
        v = self.b[self.o]
        self.o += 1
        return v

    def uint7(self):
        v = 0
        while True:
            c = self.u8()
            v = (v << 7) | (c & 0x7F)
            if not c & 0x80:
                return v

    def take(self, n):
        if n > len(self.b) - self.o:
            raise ValueError("rANS Nx16: truncated")
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def alphabet(self):
        syms, sym, run = [], self.u8(), 0
        last = sym
        while True:
            syms.append(sym)
            if run:
                run -= 1
                sym += 1
            else:
                sym = self.u8()
                if sym == last + 1:
                    run = self.u8()
            last = sym
            if sym == 0:
                return syms


def _nx16_table(freqs, bits):
    """{symbol: f} -> (F, C, slot -> symbol) with the frequencies scaled up to 2^bits; None for an all-zero row"""
    tot = sum(freqs.values())
    if tot == 0:
        return None
    up = 0
    while (tot << up) < (1 << bits):
        up += 1
    if (tot << up) != (1 << bits):
        raise ValueError("rANS Nx16: frequencies do not sum to a power of two")
    F, C, lut, acc = {}, {}, [], 0
    for sy in sorted(freqs):
        F[sy], C[sy] = freqs[sy] << up, acc
        lut += [sy] * F[sy]
        acc += F[sy]
    return F, C, lut


class _Nx16States:
    def __init__(self, src, n_states):
        self.src = src
        self.R = list(struct.unpack_from(f"<{n_states}I", src.b, src.o))
        src.o += 4 * n_states

    def symbol(self, j, table, bits):
        F, C, lut = table
        r = self.R[j]
        slot = r & ((1 << bits) - 1)
        sy = lut[slot]
        r = F[sy] * (r >> bits) + slot - C[sy]
        if r < (1 << 15):
            lo, hi = self.src.take(2)
            r = (r << 16) | lo | (hi << 8)
        self.R[j] = r
        return sy

    def finish(self):
        if any(r != (1 << 15) for r in self.R):
            raise ValueError("rANS Nx16: a state does not end at 2^15")


def _nx16_order0(src, n, n_states):
    if n == 0:
        return b""
    syms = src.alphabet()
    table = _nx16_table({sy: src.uint7() for sy in sorted(set(syms))}, 12)
    st = _Nx16States(src, n_states)
    out = bytes(st.symbol(i % n_states, table, 12) for i in range(n))
    st.finish()
    return out


def _nx16_order1(src, n, n_states):
    if n == 0:
        return b""
    comp = src.u8()
    bits, tsrc = comp >> 4, src
    if comp & 1:  # the table itself is order-0 coded, always with four states
        usz, csz = src.uint7(), src.uint7()
        tsrc = _Bytes(_nx16_order0(_Bytes(src.take(csz)), usz, 4))
    syms = sorted(set(tsrc.alphabet()))
    tables = {}
    for ctx in syms:
        row, run = {}, 0
        for sy in syms:
            if run:
                run -= 1
                row[sy] = 0
                continue
            row[sy] = tsrc.uint7()
            if row[sy] == 0:
                run = tsrc.u8()
        tables[ctx] = _nx16_table(row, bits)
    st = _Nx16States(src, n_states)
    seg = n // n_states
    out = bytearray(n)
    for j in range(n_states):  # a state never reads another one's symbols; only the 16-bit words interleave
        pass
    prev = [0] * n_states
    for k in range(seg):
        for j in range(n_states):
            prev[j] = out[j * seg + k] = st.symbol(j, tables[prev[j]], bits)
    for i in range(seg * n_states, n):
        prev[-1] = out[i] = st.symbol(n_states - 1, tables[prev[-1]], bits)
    st.finish()
    return bytes(out)


def _rans_nx16(b, expect, nested=False):
    src = _Bytes(bytes(b))
    flags = src.u8()
    n = expect if flags & 0x10 else src.uint7()
    n_states = 32 if flags & 0x04 else 4
    if flags & 0x08:
        assert not nested, "stripes inside stripes"
        x = src.u8()
        clen = [src.uint7() for _ in range(x)]
        out = bytearray(n)
        for j in range(x):
            share = n // x + (1 if n % x > j else 0)
            part = _rans_nx16(src.take(clen[j]), share, nested=True)
            assert len(part) == share
            out[j::x] = part
        return bytes(out)
    pack = None
    if flags & 0x80:
        nsym = src.u8()
        pack = (bytes(src.u8() for _ in range(nsym)), n)
        n = src.uint7()
    rle = None
    if flags & 0x40:
        m2, wide = src.uint7(), n
        n = src.uint7()
        if m2 & 1:
            meta = src.take(m2 // 2)
        else:
            csz = src.uint7()
            meta = _nx16_order0(_Bytes(src.take(csz)), m2 // 2, n_states)
        rle = (meta, wide)
    if flags & 0x20:
        data = src.take(n)
    elif flags & 0x01:
        data = _nx16_order1(src, n, n_states)
    else:
        data = _nx16_order0(src, n, n_states)
    if rle:
        meta, wide = rle
        m = _Bytes(meta)
        out = bytearray()
        if wide:
            k = m.u8() or 256
            runs = set(m.u8() for _ in range(k))
            for c in data:
                out += bytes([c]) * ((m.uint7() + 1) if c in runs else 1)
        if len(out) != wide:
            raise ValueError("rANS Nx16: runs do not fill the block")
        data = bytes(out)
    if pack:
        pmap, wide = pack
        if len(pmap) <= 1:
            data = bytes(pmap[:1]) * wide if wide else b""
        else:
            width = 1 if len(pmap) <= 2 else 2 if len(pmap) <= 4 else 4
            per, mask = 8 // width, (1 << width) - 1
            data = bytes(pmap[(data[i // per] >> ((i % per) * width)) & mask] for i in range(wide))
    return data


def _cram_block(b, o):
    method, ctype = b[o], b[o + 1]
    o += 2
    cid, o = _itf8(b, o)
    csz, o = _itf8(b, o)
    rsz, o = _itf8(b, o)
    raw = bytes(b[o:o + csz])
    if method == 1:
        raw = gzip.decompress(raw)
    elif method == 2:
        import bz2
        raw = bz2.decompress(raw)
    elif method == 3:
        import lzma
        raw = lzma.decompress(raw)
    elif method == 4:
        raw = _rans_4x8(raw)
    elif method == 5:
        raw = _rans_nx16(raw, rsz)
    elif method in (6, 7, 8):  # arithmetic coder / fqzcomp / name tokeniser: an error only if a series of the block is read
        return dict(type=ctype, id=cid, data=None, method=method), o + csz + 4
    elif method != 0:
        raise ValueError(f"CRAM block compression method {method} is not supported")
    assert len(raw) == rsz
    return dict(type=ctype, id=cid, data=raw), o + csz + 4


class _Bits:
    def __init__(self, data):
        self.d, self.p = data, 0

    def get(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | ((self.d[self.p >> 3] >> (7 - (self.p & 7))) & 1)
            self.p += 1
        return v


def _cram_encoding(b, o):
    codec, o = _itf8(b, o)
    ln, o = _itf8(b, o)
    p, e = o, o + ln
    if codec == 0:
        enc = ("null",)
    elif codec == 1:
        cid, p = _itf8(b, p)
        enc = ("external", cid)
    elif codec == 3:
        n, p = _itf8(b, p)
        syms = []
        for _ in range(n):
            s, p = _itf8s(b, p)
            syms.append(s)
        n2, p = _itf8(b, p)
        lens = []
        for _ in range(n2):
            l, p = _itf8(b, p)
            lens.append(l)
        enc = ("huffman", syms, lens)
    elif codec == 4:
        le, p = _cram_encoding(b, p)
        ve, p = _cram_encoding(b, p)
        enc = ("byte_array_len", le, ve)
    elif codec == 5:
        stop = b[p]
        cid, p = _itf8(b, p + 1)
        enc = ("byte_array_stop", stop, cid)
    elif codec == 6:
        off, p = _itf8s(b, p)
        nbits, p = _itf8(b, p)
        enc = ("beta", off, nbits)
    else:
        raise ValueError(f"CRAM encoding {codec} is not supported")
    return enc, e


class _Slice:
    def __init__(self, core, ext):
        self.core, self.ext, self.pos = _Bits(core), ext, {k: 0 for k in ext}

    def int(self, enc):
        k = enc[0]
        if k == "external":
            if self.ext[enc[1]] is None:
                raise ValueError("an integer data series lives in a block whose compression method is not supported")
            v, self.pos[enc[1]] = _itf8s(self.ext[enc[1]], self.pos[enc[1]])
            return v
        if k == "huffman":
            syms, lens = enc[1], enc[2]
            if len(syms) == 1 and lens[0] == 0:
                return syms[0]
            order = sorted(range(len(syms)), key=lambda i: (lens[i], syms[i]))  # canonical codes
            code, prev_len, codes = 0, 0, {}
            for i in order:
                code <<= lens[i] - prev_len
                codes[(lens[i], code)] = syms[i]
                prev_len = lens[i]
                code += 1
            v, n = 0, 0
            while True:
                v = (v << 1) | self.core.get(1)
                n += 1
                if (n, v) in codes:
                    return codes[(n, v)]
        if k == "beta":
            return self.core.get(enc[2]) - enc[1]
        raise ValueError(f"integer through {k}")

    # a block coded with fqzcomp / the name tokeniser / the arithmetic coder has data None (_cram_block): the byte series in it
    # (quality scores, names -- where CRAM 3.1 writers use those codecs) read as None / 0; an integer series in one is an error
    def byte(self, enc):
        if enc[0] == "external":
            if self.ext[enc[1]] is None:
                return 0
            v = self.ext[enc[1]][self.pos[enc[1]]]
            self.pos[enc[1]] += 1
            return v
        return self.int(enc) & 0xFF

    def bytes(self, enc):
        if enc[0] == "byte_array_stop":
            d, p = self.ext[enc[2]], self.pos[enc[2]]
            if d is None:
                return None
            e = d.index(bytes([enc[1]]), p)
            self.pos[enc[2]] = e + 1
            return d[p:e]
        if enc[0] == "byte_array_len":
            n = self.int(enc[1])
            if enc[2][0] == "external":
                p = self.pos[enc[2][1]]
                self.pos[enc[2][1]] = p + n
                if self.ext[enc[2][1]] is None:
                    return None
                return self.ext[enc[2][1]][p:p + n]
            return bytes(self.byte(enc[2]) for _ in range(n))
        raise ValueError(f"byte array through {enc[0]}")


def decode_cram(path):
    """-> (references [(name, length)], records [dict(name, flag, ref_id, start, end, mapq, cigar)]).  Columns as
    exon-cram/src/array_builder.rs: reference id -1 -> None, start = alignment position (0 -> None), end = start + reference
    span - 1 with the span from the read length and the read features, mapping quality 255 -> None."""
    b = read_bytes(path)
    assert b[:4] == b"CRAM" and b[4] == 3, "CRAM 3.x expected"
    o, refs, recs, first = 26, [], [], True
    while o < len(b):
        c_off = o  # where this container's header starts (what a .crai entry's fourth field holds)
        length, = struct.unpack_from("<i", b, o)
        o += 4
        ref_id, o = _itf8s(b, o)
        _start, o = _itf8(b, o)
        _span, o = _itf8(b, o)
        n_rec, o = _itf8(b, o)
        _rc, o = _ltf8(b, o)
        _bases, o = _ltf8(b, o)
        n_blocks, o = _itf8(b, o)
        n_land, o = _itf8(b, o)
        for _ in range(n_land):
            _, o = _itf8(b, o)
        o += 4
        end = o + length
        if first:  # the SAM header container
            blk, _ = _cram_block(b, o)
            l_text, = struct.unpack_from("<i", blk["data"], 0)
            for line in blk["data"][4:4 + l_text].decode().split("\n"):
                if line.startswith("@SQ"):
                    f = dict(x.split(":", 1) for x in line.split("\t")[1:] if ":" in x)
                    refs.append((f["SN"], int(f.get("LN", 0))))
            first, o = False, end
            continue
        if n_rec == 0 and ref_id == -1 and n_blocks <= 1:  # EOF container
            o = end
            continue
        blk, p = _cram_block(b, o)
        assert blk["type"] == 1
        h, q = blk["data"], 0
        # preservation map
        _sz, q = _itf8(h, q)
        n, q = _itf8(h, q)
        pres = dict(RN=True, AP=True, RR=True)
        tag_lines = []
        for _ in range(n):
            key = h[q:q + 2].decode()
            q += 2
            if key in ("RN", "AP", "RR"):
                pres[key] = h[q] != 0
                q += 1
            elif key == "SM":
                q += 5
            elif key == "TD":
                ln, q = _itf8(h, q)
                for line in h[q:q + ln].split(b"\0")[:-1]:
                    tag_lines.append([line[i:i + 3] for i in range(0, len(line), 3)])
                q += ln
            else:
                raise ValueError(f"preservation key {key}")
        # data series encodings
        _sz, q = _itf8(h, q)
        n, q = _itf8(h, q)
        ds = {}
        for _ in range(n):
            key = h[q:q + 2].decode()
            ds[key], q = _cram_encoding(h, q + 2)
        # tag encodings
        _sz, q = _itf8(h, q)
        n, q = _itf8(h, q)
        tags = {}
        for _ in range(n):
            key, q = _itf8(h, q)
            tags[key], q = _cram_encoding(h, q)
        # slices
        while p < end:
            s_off = p - (end - length)  # the slice header's offset from the end of the container header (.crai field five)
            sh, p = _cram_block(b, p)
            assert sh["type"] == 2
            d, q = sh["data"], 0
            s_ref, q = _itf8s(d, q)
            s_start, q = _itf8(d, q)
            _s_span, q = _itf8(d, q)
            s_nrec, q = _itf8(d, q)
            _ctr, q = _ltf8(d, q)
            s_nblocks, q = _itf8(d, q)
            core, ext = b"", {}
            for _ in range(s_nblocks):
                blk, p = _cram_block(b, p)
                if blk["type"] == 5:
                    core = blk["data"]
                elif blk["type"] == 4:
                    ext[blk["id"]] = blk["data"]
            sl = _Slice(core, ext)
            prev = s_start
            for _ in range(s_nrec):
                bf, cf = sl.int(ds["BF"]), sl.int(ds["CF"])
                ri = sl.int(ds["RI"]) if s_ref == -2 else s_ref
                rl = sl.int(ds["RL"])
                ap = sl.int(ds["AP"])
                if pres["AP"]:
                    ap += prev
                    prev = ap
                sl.int(ds["RG"])
                name = sl.bytes(ds["RN"]) if pres["RN"] else b""
                if cf & 2:
                    sl.int(ds["MF"])
                    if not pres["RN"]:
                        name = sl.bytes(ds["RN"])
                    sl.int(ds["NS"]); sl.int(ds["NP"]); sl.int(ds["TS"])
                elif cf & 4:
                    sl.int(ds["NF"])
                tl = sl.int(ds["TL"])
                for t in tag_lines[tl]:
                    sl.bytes(tags[(t[0] << 16) | (t[1] << 8) | t[2]])
                mapq, span, cigar = 255, rl, None
                if not bf & 4:
                    fn = sl.int(ds["FN"])
                    feats, at = [], 0
                    for _ in range(fn):
                        code = chr(sl.byte(ds["FC"]))
                        at += sl.int(ds["FP"])
                        if code == "B":
                            sl.byte(ds["BA"]); sl.byte(ds["QS"])
                        elif code == "X":
                            sl.byte(ds["BS"])
                        elif code == "I":
                            n_ins = len(sl.bytes(ds["IN"])); span -= n_ins; feats.append((at, "I", n_ins))
                        elif code == "i":
                            sl.byte(ds["BA"]); span -= 1; feats.append((at, "I", 1))
                        elif code == "D":
                            n_del = sl.int(ds["DL"]); span += n_del; feats.append((at, "D", n_del))
                        elif code == "S":
                            n_sc = len(sl.bytes(ds["SC"])); span -= n_sc; feats.append((at, "S", n_sc))
                        elif code == "N":
                            n_sk = sl.int(ds["RS"]); span += n_sk; feats.append((at, "N", n_sk))
                        elif code == "P":
                            feats.append((at, "P", sl.int(ds["PD"])))
                        elif code == "H":
                            feats.append((at, "H", sl.int(ds["HC"])))
                        elif code == "Q":
                            sl.byte(ds["QS"])
                        elif code == "b":
                            sl.bytes(ds["BB"])
                        elif code == "q":
                            sl.bytes(ds["QQ"])
                        else:
                            raise ValueError(f"read feature {code!r}")
                    mapq = sl.int(ds["MQ"])
                    if cf & 1:
                        for _ in range(rl):
                            sl.byte(ds["QS"])
                    # CIGAR from the features: matches fill the gaps between the features that consume read bases
                    ops, rp = [], 1
                    for at, op, n_op in feats:
                        if at > rp:
                            ops.append((at - rp, "M")); rp = at
                        ops.append((n_op, op))
                        if op in "IS":
                            rp += n_op
                    if rp <= rl:
                        ops.append((rl - rp + 1, "M"))
                    merged = []
                    for n_op, op in ops:
                        if merged and merged[-1][1] == op:
                            merged[-1] = (merged[-1][0] + n_op, op)
                        else:
                            merged.append((n_op, op))
                    cigar = "".join(f"{n_op}{op}" for n_op, op in merged)
                else:
                    for _ in range(rl):
                        sl.byte(ds["BA"])
                    if cf & 1:
                        for _ in range(rl):
                            sl.byte(ds["QS"])
                start = ap if ap >= 1 else None
                recs.append(dict(name=None if name is None else name.decode(), container=c_off, slice=s_off, flag=bf, ref_id=ri if ri >= 0 else None, start=start,
                                 end=(start + (0 if bf & 4 else span) - 1) if start is not None else None,
                                 mapq=None if mapq == 255 else mapq, cigar=cigar))
        o = end
    return refs, recs
