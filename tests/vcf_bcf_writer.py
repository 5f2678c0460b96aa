"""Test-side writer of a VCF text file and its BCF 2.2 twin with typed INFO fields of every kind the reference builds
(exon-core/src/datasources/vcf/schema_builder.rs:197-249): Float / Integer scalars, a Flag, a String scalar, and lists
(Number=A Integer, Number=. Float, Number=. String).  Independent of the product's decoders and of the oracle: rows are
python dicts, the BCF side encodes typed values the way htslib does (smallest integer type that holds every item)."""
import struct
import subprocess

import numpy as np

INFO_HEADER = [("AF", "1", "Float"), ("DP", "1", "Integer"), ("DB", "0", "Flag"), ("CSQ", "1", "String"),
               ("AC", "A", "Integer"), ("MQS", ".", "Float"), ("TAGS", ".", "String")]
FILTERS = ["q10", "s50"]


def header_text(bcf):
    idx = 0
    lines = ["##fileformat=VCFv4.3", '##FILTER=<ID=PASS,Description="All filters passed"' + (",IDX=0>" if bcf else ">"),
             "##contig=<ID=1" + (",IDX=0>" if bcf else ">"), "##contig=<ID=2" + (",IDX=1>" if bcf else ">")]
    for f in FILTERS:
        idx += 1
        lines.append(f'##FILTER=<ID={f},Description="x"' + (f",IDX={idx}>" if bcf else ">"))
    for name, number, typ in INFO_HEADER:
        idx += 1
        lines.append(f'##INFO=<ID={name},Number={number},Type={typ},Description="x"' + (f",IDX={idx}>" if bcf else ">"))
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO")
    return "\n".join(lines) + "\n"


def string_index():
    names = ["PASS"] + FILTERS + [n for n, _, _ in INFO_HEADER]
    return {n: i for i, n in enumerate(names)}


def make_rows(n, seed=7):
    """rows: dict(chrom, pos, qual (float | None), filter (list[str]), info (dict | None)).  info values: float, int, True,
    str, or lists with None items; a key mapped to None means `key=.`"""
    rng = np.random.default_rng(seed)
    edge_dp = [16777215, 16777216, 16777217, 16777218, 16777219, 2**31 - 1, -(2**31) + 8, -5, 0, 127, 128, -120, 32767, 32768, 100000]
    rows = []
    for i in range(n):
        info = {}
        if rng.random() < 0.9:
            info["AF"] = float(np.float32(rng.choice([0.001, 0.01, 0.0100000001, 0.25, 0.5, 1e-5, 3.0e-2])))
        if rng.random() < 0.85:
            info["DP"] = int(edge_dp[i % len(edge_dp)]) if rng.random() < 0.5 else int(rng.integers(0, 200))
        elif rng.random() < 0.3:
            info["DP"] = None  # DP=.
        if rng.random() < 0.3:
            info["DB"] = True
        if rng.random() < 0.4:
            info["CSQ"] = str(rng.choice(["missense", "stop", "syn"]))
        if rng.random() < 0.6:
            k = int(rng.integers(1, 4))
            info["AC"] = [None if rng.random() < 0.15 else int(rng.choice([1, 2, 300, 70000, -3])) for _ in range(k)]
        if rng.random() < 0.4:
            k = int(rng.integers(1, 5))
            info["MQS"] = [None if rng.random() < 0.1 else float(np.float32(rng.choice([60.0, 37.5, 0.125, 1e-3]))) for _ in range(k)]
        if rng.random() < 0.3:
            k = int(rng.integers(1, 3))
            info["TAGS"] = [None if rng.random() < 0.1 else str(rng.choice(["a", "bb", "ccc"])) for _ in range(k)]
        for k in ("AC", "MQS", "TAGS"):  # a one-item list whose item is missing IS `key=.`: the whole value is missing (NULL list)
            if info.get(k) == [None]:
                info[k] = None
        filt = [[], ["PASS"], ["q10"], ["q10", "s50"], ["s50"]][int(rng.integers(0, 5))]
        qual = None if rng.random() < 0.05 else float(np.float32(int(rng.integers(0, 10000)) / 10))
        rows.append(dict(chrom=str(1 + (i >= n // 2)), pos=i + 1, qual=qual, filter=filt, info=(info if info and rng.random() < 0.97 else None)))
    return rows


def _info_text(info):
    if info is None:
        return "."
    parts = []
    for k, v in info.items():
        if v is True:
            parts.append(k)
        elif v is None:
            parts.append(f"{k}=.")
        elif isinstance(v, list):
            parts.append(k + "=" + ",".join("." if e is None else (np.format_float_positional(np.float32(e), unique=True, trim="0")
                                                                    if isinstance(e, float) else str(e)) for e in v))
        elif isinstance(v, float):
            parts.append(f"{k}={np.format_float_positional(np.float32(v), unique=True, trim='0')}")
        else:
            parts.append(f"{k}={v}")
    return ";".join(parts)


def write_vcf(path, rows):
    with open(path, "w") as f:
        f.write(header_text(False))
        for r in rows:
            q = "." if r["qual"] is None else np.format_float_positional(np.float32(r["qual"]), unique=True, trim="0")
            f.write(f"{r['chrom']}\t{r['pos']}\t.\tA\tC\t{q}\t{';'.join(r['filter']) or '.'}\t{_info_text(r['info'])}\n")


def _typed_ints(vals):
    """typed integer vector (None = missing) in the smallest type that holds every item (reserving the 8 lowest values)"""
    present = [v for v in vals if v is not None]
    lo, hi = (min(present), max(present)) if present else (0, 0)
    if -120 <= lo and hi <= 127:
        t, fmt, miss = 1, "b", -128
    elif -32760 <= lo and hi <= 32767:
        t, fmt, miss = 2, "h", -32768
    else:
        t, fmt, miss = 3, "i", -2147483648
    return _desc(len(vals), t) + b"".join(struct.pack("<" + fmt, miss if v is None else v) for v in vals)


def _desc(n, t):
    if n < 15:
        return bytes([(n << 4) | t])
    return bytes([0xF0 | t]) + _typed_ints([n])


def _typed_floats(vals):
    return _desc(len(vals), 5) + b"".join(struct.pack("<I", 0x7F800001) if v is None else struct.pack("<f", np.float32(v)) for v in vals)


def _typed_str(s):
    b = s.encode()
    return _desc(len(b), 7) + b


def write_bcf(path, rows, bgzip):
    """uncompressed BCF stream -> `bgzip` (tools/bin/bgzip) -> path"""
    sidx = string_index()
    types = {n: (num, typ) for n, num, typ in INFO_HEADER}
    text = header_text(True).encode() + b"\0"
    out = [b"BCF\x02\x02", struct.pack("<I", len(text)), text]
    for r in rows:
        info = r["info"] or {}
        shared = struct.pack("<iiiIII", int(r["chrom"]) - 1, r["pos"] - 1, 1,
                             0x7F800001 if r["qual"] is None else struct.unpack("<I", struct.pack("<f", np.float32(r["qual"])))[0],
                             len(info) | (2 << 16), 0)
        shared += b"\x07" + _typed_str("A") + _typed_str("C")
        shared += _typed_ints([sidx[f] for f in r["filter"]]) if r["filter"] else b"\x00"
        for k, v in info.items():
            shared += _typed_ints([sidx[k]])
            num, typ = types[k]
            if typ == "Flag":
                shared += b"\x00"
            elif v is None:
                shared += (_typed_floats([None]) if typ == "Float" else _typed_ints([None]) if typ == "Integer" else _typed_str("."))
            elif typ == "Integer":
                shared += _typed_ints(v if isinstance(v, list) else [v])
            elif typ == "Float":
                shared += _typed_floats(v if isinstance(v, list) else [v])
            else:
                shared += _typed_str(",".join("." if e is None else e for e in v) if isinstance(v, list) else v)
        out.append(struct.pack("<II", len(shared), 0) + shared)
    raw = str(path) + ".u"
    with open(raw, "wb") as f:
        f.write(b"".join(out))
    subprocess.check_call([bgzip, raw, str(path), "6"])


def expected_column(rows, key):
    """what info.<key> must decode to (python values; floats f32-rounded), straight from the rows"""
    num, typ = {n: (a, b) for n, a, b in INFO_HEADER}[key]
    out = []
    for r in rows:
        v = None if r["info"] is None else r["info"].get(key)
        if typ == "Flag":
            out.append(True if v else None)
        elif isinstance(v, list):
            out.append([None if e is None else (float(np.float32(e)) if typ == "Float" else e) for e in v])
        elif isinstance(v, float):
            out.append(float(np.float32(v)))
        else:
            out.append(v)
    return out
