"""Test infrastructure: a small CRAM 3.0 WRITER, so that the CRAM front end can be checked on files of our own -- many
containers and slices, single- and multi-reference slices, absolute and delta positions, detached mates, every read feature
that changes the reference span -- against the truth the writer started from.  (The image has no htslib; the reference's
fixtures are 2 to 910 records.)  Raw and gzip blocks; integer series as EXTERNAL (ITF8), BETA or multi-symbol HUFFMAN in the
core bit stream; names BYTE_ARRAY_STOP, insertions / soft clips BYTE_ARRAY_LEN.  Not product code and not the oracle."""
import gzip
import struct
import zlib
import bz2
import lzma

import numpy as np


def itf8(v):
    v &= 0xFFFFFFFF
    if v < 0x80:
        return bytes([v])
    if v < 0x4000:
        return bytes([0x80 | (v >> 8), v & 0xFF])
    if v < 0x200000:
        return bytes([0xC0 | (v >> 16), (v >> 8) & 0xFF, v & 0xFF])
    if v < 0x10000000:
        return bytes([0xE0 | (v >> 24), (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])
    return bytes([0xF0 | (v >> 28), (v >> 20) & 0xFF, (v >> 12) & 0xFF, (v >> 4) & 0xFF, v & 0x0F])


def ltf8(v):
    assert 0 <= v < (1 << 49)
    if v < 0x80:
        return bytes([v])
    for n in range(1, 7):  # n extra bytes: 7 + 7n ... bits
        if v < (1 << (7 - n + 8 * n)):
            head = (0xFF << (8 - n)) & 0xFF
            return bytes([head | (v >> (8 * n))]) + bytes((v >> (8 * (n - 1 - i))) & 0xFF for i in range(n))
    raise ValueError(v)


def block(ctype, cid, data, method=0, nx16=None, payload=None):
    """method: 0 raw, 1 gzip, 2 bzip2, 3 lzma (an .xz stream) -- CRAM 3.0 section 8.1; 5 rANS Nx16 (CRAM 3.1; `nx16` = keyword
    arguments of rans_nx16_writer.encode: flags ...).  `payload`: bytes to store as they are under `method` (a block coded with
    a method nobody here can produce -- fqzcomp, name tokeniser: a reader that needs no series of it must never open it)"""
    if payload is not None:
        comp = payload
    elif method == 5:
        import rans_nx16_writer
        comp = rans_nx16_writer.encode(data, **(nx16 or {}))
    else:
        comp = {0: lambda d: d, 1: lambda d: gzip.compress(d, mtime=0), 2: bz2.compress, 3: lzma.compress}[method](data)
    body = bytes([method, ctype]) + itf8(cid) + itf8(len(comp)) + itf8(len(data)) + comp
    return body + struct.pack("<I", zlib.crc32(body) & 0xFFFFFFFF)


def container(ref, start, span, nrec, counter, bases, blocks, landmarks):
    payload = b"".join(blocks)
    head = struct.pack("<i", len(payload)) + itf8(ref) + itf8(start) + itf8(span) + itf8(nrec) + ltf8(counter) + ltf8(bases) + \
        itf8(len(blocks)) + itf8(len(landmarks)) + b"".join(itf8(x) for x in landmarks)
    return head + struct.pack("<I", zlib.crc32(head) & 0xFFFFFFFF) + payload


class Bits:
    def __init__(self):
        self.bits = []

    def put(self, v, n):
        self.bits += [(v >> (n - 1 - i)) & 1 for i in range(n)]

    def bytes(self):
        b = self.bits + [0] * (-len(self.bits) % 8)
        return bytes(sum(b[i + j] << (7 - j) for j in range(8)) for i in range(0, len(b), 8))


def enc_external(cid):
    return itf8(1) + itf8(len(itf8(cid))) + itf8(cid)


def enc_beta(offset, nbits):
    p = itf8(offset) + itf8(nbits)
    return itf8(6) + itf8(len(p)) + p


def enc_huffman(syms, lens):
    p = itf8(len(syms)) + b"".join(itf8(s) for s in syms) + itf8(len(lens)) + b"".join(itf8(l) for l in lens)
    return itf8(3) + itf8(len(p)) + p


def enc_stop(stop, cid):
    p = bytes([stop]) + itf8(cid)
    return itf8(5) + itf8(len(p)) + p


def enc_len(len_enc, val_enc):
    p = len_enc + val_enc
    return itf8(4) + itf8(len(p)) + p


def canonical(syms, lens):
    order = sorted(range(len(syms)), key=lambda i: (lens[i], syms[i]))
    code, prev, out = 0, 0, {}
    for i in order:
        code <<= lens[i] - prev
        out[syms[i]] = (code, lens[i])
        prev = lens[i]
        code += 1
    return out


MQ_SYMS, MQ_LENS = [60, 0, 30, 255, 13], [1, 2, 3, 4, 4]  # a complete prefix code in the core bit stream
IDS = dict(BF=1, CF=2, RI=3, RL=4, AP=5, RG=6, RN=7, MF=8, NS=9, NP=10, TS=11, NF=12, TL=13, FN=14, FC=15, FP=16, DL=17, BA=18,
           QS=19, BS=20, IN=21, SC=22, RS=23, PD=24, HC=25, LEN=26)


NX16_FLAGS = (0x00, 0x01, 0x04, 0x05, 0x40, 0x41, 0x80, 0x81, 0xC0, 0xC1, 0xC5, 0x08, 0x09, 0x0C, 0x20)  # what a 3.1 writer may pick per block


def write_cram(path, refs, records, per_slice=700, slices_per_container=2, seed=0, ds_patch=None, methods=(0, 1), qualities=False,
               opaque=None):
    """refs: [(name, length)]; records: dicts(flag, ref_id (-1 unmapped), pos (1-based, 0 none), mapq, name, rl, feats) with
    feats = [(read position, code, value)], code in I i D S N P H X.  Records are written in the given order; a run of records
    on one reference makes single-reference slices, mixed runs make multi-reference (-2) slices.  `methods`: the block
    compression methods the external blocks draw from (0 raw, 1 gzip, 2 bzip2, 3 lzma).  `qualities`: records carry their
    quality scores (CF bit 0, `rl` bytes in the QS series) -- most of a real file's bytes.  Method 5 (rANS Nx16) makes the file
    CRAM 3.1, each such block with flags drawn from NX16_FLAGS.  `opaque`: {data series: method} -- the series' block is
    written under that method number with a payload no decoder understands (stands for fqzcomp / name-tokeniser blocks)."""
    opaque = {IDS[k]: m for k, m in (opaque or {}).items()}
    v31 = any(m >= 5 for m in methods) or bool(opaque)

    def ext_block(k, data):
        if k in opaque:
            return block(4, k, data, opaque[k], payload=bytes(reversed(data)) + b"?")
        m = methods[int(rng.integers(0, len(methods)))]
        if m == 5:
            nx16 = dict(flags=NX16_FLAGS[int(rng.integers(0, len(NX16_FLAGS)))], o1_bits=(10, 12)[int(rng.integers(0, 2))],
                        code_table=bool(rng.integers(0, 2)), code_rle_meta=bool(rng.integers(0, 2)))
            return block(4, k, data, 5, nx16=nx16)
        return block(4, k, data, m)
    rng = np.random.default_rng(seed)
    text = "@HD\tVN:1.6\tSO:unsorted\n" + "".join(f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in refs)
    hdr = struct.pack("<i", len(text)) + text.encode()
    out = [(b"CRAM\x03\x01" if v31 else b"CRAM\x03\x00") + b"exon-hip-test-file\0\0", container(0, 0, 0, 0, 0, 0, [block(0, 0, hdr, 1)], [0])]
    mq = canonical(MQ_SYMS, MQ_LENS)
    counter = 0
    groups = [records[i:i + per_slice] for i in range(0, len(records), per_slice)]
    for ci in range(0, len(groups), slices_per_container):
        cont = groups[ci:ci + slices_per_container]
        ap_delta = bool((ci // slices_per_container) & 1)
        rn_kept = True
        pres = [b"RN" + bytes([1 if rn_kept else 0]), b"AP" + bytes([1 if ap_delta else 0]), b"RR\x00",
                b"TD" + itf8(1) + b"\0"]
        pmap = itf8(len(pres)) + b"".join(pres)
        ds = {k: enc_external(IDS[k]) for k in ("BF", "CF", "RI", "AP", "RG", "MF", "NS", "NP", "TS", "NF", "FN", "FC", "FP", "DL", "BA",
                                                "QS", "BS", "RS", "PD", "HC")}
        ds["RL"] = enc_beta(0, 9)                       # core bits
        ds["TL"] = enc_huffman([0], [0])                # a single symbol: no bits
        ds["MQ"] = enc_huffman(MQ_SYMS, MQ_LENS)        # core bits
        ds["RN"] = enc_stop(0, IDS["RN"])
        ds["IN"] = enc_stop(9, IDS["IN"])
        ds["SC"] = enc_len(enc_external(IDS["LEN"]), enc_external(IDS["SC"]))
        ds.update(ds_patch or {})  # tests: replace / add data-series encodings (raw bytes: codec id, length, parameters)
        dmap = itf8(len(ds)) + b"".join(k.encode() + v for k, v in ds.items())
        tmap = itf8(0)
        ch = block(1, 0, itf8(len(pmap)) + pmap + itf8(len(dmap)) + dmap + itf8(len(tmap)) + tmap)
        blocks, landmarks, off = [ch], [], len(ch)
        c_refs, c_lo, c_hi, c_n, c_bases = set(), None, 0, 0, 0
        for sl in cont:
            ext = {k: bytearray() for k in IDS.values()}
            core = Bits()
            rset = {r["ref_id"] for r in sl}
            s_ref = rset.pop() if len(rset) == 1 else -2
            mapped = [r for r in sl if r["pos"] > 0]
            s_start = min((r["pos"] for r in mapped), default=0) if s_ref >= 0 else 0
            s_end = max((r["pos"] + r["span"] - 1 for r in mapped), default=0) if s_ref >= 0 else 0
            prev = s_start
            for r in sl:
                cf = (2 if r.get("detached") else 0) | (1 if qualities else 0)
                ext[IDS["BF"]] += itf8(r["flag"])
                ext[IDS["CF"]] += itf8(cf)
                if s_ref == -2:
                    ext[IDS["RI"]] += itf8(r["ref_id"])
                core.put(r["rl"], 9)
                ext[IDS["AP"]] += itf8(r["pos"] - prev if ap_delta else r["pos"])
                if ap_delta:
                    prev = r["pos"]
                ext[IDS["RG"]] += itf8(-1)
                ext[IDS["RN"]] += r["name"].encode() + b"\0"
                if cf & 2:
                    ext[IDS["MF"]] += itf8(1)
                    ext[IDS["NS"]] += itf8(r["ref_id"]); ext[IDS["NP"]] += itf8(r["pos"] + 200); ext[IDS["TS"]] += itf8(300)
                if not r["flag"] & 4:
                    ext[IDS["FN"]] += itf8(len(r["feats"]))
                    at = 0
                    for pos, code, val in r["feats"]:
                        ext[IDS["FC"]] += code.encode()
                        ext[IDS["FP"]] += itf8(pos - at)
                        at = pos
                        if code == "I":
                            ext[IDS["IN"]] += b"A" * val + b"\x09"
                        elif code == "i":
                            ext[IDS["BA"]] += b"C"
                        elif code == "D":
                            ext[IDS["DL"]] += itf8(val)
                        elif code == "S":
                            ext[IDS["LEN"]] += itf8(val); ext[IDS["SC"]] += b"G" * val
                        elif code == "N":
                            ext[IDS["RS"]] += itf8(val)
                        elif code == "P":
                            ext[IDS["PD"]] += itf8(val)
                        elif code == "H":
                            ext[IDS["HC"]] += itf8(val)
                        elif code == "X":
                            ext[IDS["BS"]] += bytes([1])
                    code, nb = mq[r["mapq"]]
                    core.put(code, nb)
                else:
                    ext[IDS["BA"]] += b"N" * r["rl"]
                if qualities:
                    ext[IDS["QS"]] += rng.integers(2, 42, r["rl"], dtype=np.uint8).tobytes()
                c_bases += r["rl"]
            used = [k for k, v in ext.items() if v]
            body = itf8(s_ref) + itf8(s_start) + itf8(max(0, s_end - s_start + 1) if s_ref >= 0 else 0) + itf8(len(sl)) + ltf8(counter) + \
                itf8(1 + len(used)) + itf8(1 + len(used)) + itf8(0) + b"".join(itf8(k) for k in used) + itf8(-1) + bytes(16)
            sh = block(2, 0, body)
            landmarks.append(off)
            sblocks = [sh, block(5, 0, core.bytes(), 5, nx16=dict(flags=int(rng.integers(0, 2)))) if 5 in methods else block(5, 0, core.bytes())] + [ext_block(k, bytes(ext[k])) for k in used]
            blocks += sblocks
            off += sum(len(x) for x in sblocks)
            counter += len(sl)
            c_n += len(sl)
            c_refs.add(s_ref)
            if s_ref >= 0 and mapped:
                c_lo = s_start if c_lo is None else min(c_lo, s_start)
                c_hi = max(c_hi, s_end)
        c_ref = c_refs.pop() if len(c_refs) == 1 else -2
        out.append(container(c_ref, (c_lo or 0) if c_ref >= 0 else 0, (c_hi - c_lo + 1) if c_ref >= 0 and c_lo else 0, c_n, counter - c_n,
                             c_bases, blocks, landmarks))
    out.append(container(-1, 4542278, 0, 0, 0, 0, [block(1, 0, b"\x01\x00\x01\x00\x01\x00")], []))  # EOF container
    open(path, "wb").write(b"".join(out))


def synthetic_records(n, refs, seed=1):
    """sorted-ish records with every span-changing feature; returns the dicts write_cram takes (with `span` = reference span)"""
    rng = np.random.default_rng(seed)
    recs = []
    ref_of = np.sort(rng.integers(0, len(refs), n))
    for i in range(n):
        unmapped = i % 37 == 36
        rl = int(rng.integers(20, 151))
        rid = -1 if unmapped else int(ref_of[i])
        feats, span = [], rl
        if not unmapped:
            at = 1
            for _ in range(int(rng.integers(0, 4))):
                at += int(rng.integers(1, 12))
                if at >= rl - 12:
                    break
                code = "IiDSNPHX"[int(rng.integers(0, 8))]
                val = int(rng.integers(1, 9))
                if code == "N":
                    val = int(rng.integers(50, 5000))
                if code == "I":
                    span -= val
                elif code == "i":
                    span -= 1
                elif code == "S":
                    span -= val
                elif code in "DN":
                    span += val
                feats.append((at, code, val))
                if code in "IS":
                    at += val
        pos = 0 if unmapped else int(rng.integers(1, refs[rid][1] - 6000))
        recs.append(dict(flag=4 if unmapped else int(rng.choice([0, 16, 99, 147, 83, 163, 1024 + 99])), ref_id=rid, pos=pos,
                         mapq=int(rng.choice(MQ_SYMS)) if not unmapped else 255, name=f"q{i}", rl=rl, feats=feats,
                         span=0 if unmapped else span, detached=(i % 11 == 0 and not unmapped)))
    # positions ascending inside a reference (delta coding then stays small, as in a sorted file)
    recs.sort(key=lambda r: (r["ref_id"] if r["ref_id"] >= 0 else 1 << 30, r["pos"]))
    return recs
