"""Test-side ENCODER of CRAM 3.1's rANS Nx16 block codec (method 5) -- test infrastructure, nothing here is shipped.

Written from the published "CRAM codecs" specification (section 3) so that the product's decoder (exon_amd/csrc/host/cram.h
`rans_nx16`) and the oracle's (oracle/decode.py `_rans_nx16`) have streams to decode: this image has no htslib and the reference's
fixtures are CRAM 3.0.  Every option of the flags byte can be produced: order 0 / 1, 4 or 32 states, bit-packing, run-length
coding (side stream stored or order-0 coded), striping, stored data, size omitted.

Layout of a stream:  flags  [size: uint7]  [pack: nsym, symbols, packed size]  [rle: 2*meta size (+1 = stored), literal count,
(coded meta size,) meta]  data.   uint7 = big-endian base 128, high bit "more".
"""
import struct

ORDER, X32, STRIPE, NOSZ, CAT, RLE, PACK = 0x01, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80
LOW = 1 << 15


def uint7(v):
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append(0x80 | (v & 0x7F))
        v >>= 7
    return bytes(reversed(out))


def alphabet(syms):
    """ascending symbols; a symbol that follows its predecessor announces how many more consecutive ones come after it"""
    syms = sorted(syms)
    present = set(syms)
    out, skip = bytearray(), 0
    for sy in syms:
        if skip:
            skip -= 1
            continue
        out.append(sy)
        if sy > 0 and sy - 1 in present:
            run = 0
            while sy + run + 1 in present:
                run += 1
            out.append(run)
            skip = run
    out.append(0)
    return bytes(out)


def normalise(counts, bits):
    """{symbol: count > 0} -> {symbol: frequency >= 1} summing to 2^bits"""
    total, full = sum(counts.values()), 1 << bits
    F = {s: max(1, c * full // total) for s, c in counts.items()}
    diff = full - sum(F.values())
    order = sorted(F, key=lambda s: -F[s])
    if diff > 0:
        F[order[0]] += diff
    while diff < 0:
        for s in order:
            if diff < 0 and F[s] > 1:
                take = min(F[s] - 1, -diff, max(1, F[s] // 2))
                F[s] -= take
                diff += take
    assert sum(F.values()) == full and min(F.values()) >= 1
    return F


def _cumulative(F):
    C, acc = {}, 0
    for s in sorted(F):
        C[s] = acc
        acc += F[s]
    return C


def _emit(steps, n_states, bits):
    """steps: (state, frequency, cumulative) in DECODING order; the encoder walks them backwards"""
    R, words = [LOW] * n_states, []
    for j, f, c in reversed(steps):
        x = R[j]
        if x >= f << (31 - bits):
            words.append(x & 0xFFFF)
            x >>= 16
        R[j] = ((x // f) << bits) + (x % f) + c
    return struct.pack(f"<{n_states}I", *R) + b"".join(struct.pack("<H", w) for w in reversed(words))


def order0(data, n_states=4):
    if not data:
        return b""
    counts = {}
    for b in data:
        counts[b] = counts.get(b, 0) + 1
    F = normalise(counts, 12)
    C = _cumulative(F)
    head = alphabet(F) + b"".join(uint7(F[s]) for s in sorted(F))
    return head + _emit([(i % n_states, F[b], C[b]) for i, b in enumerate(data)], n_states, 12)


def order1(data, n_states=4, bits=12, code_table=False):
    if not data:
        return b""
    n, seg = len(data), len(data) // n_states
    walk = []  # (state, context, symbol) in decoding order
    prev = [0] * n_states
    for k in range(seg):
        for j in range(n_states):
            b = data[j * seg + k]
            walk.append((j, prev[j], b))
            prev[j] = b
    for i in range(seg * n_states, n):
        walk.append((n_states - 1, prev[-1], data[i]))
        prev[-1] = data[i]
    counts = {}
    for _j, ctx, b in walk:
        counts.setdefault(ctx, {})
        counts[ctx][b] = counts[ctx].get(b, 0) + 1
    syms = sorted(set(data) | {0} | set(counts))
    F = {ctx: normalise(row, bits) for ctx, row in counts.items()}
    C = {ctx: _cumulative(row) for ctx, row in F.items()}
    table = bytearray(alphabet(syms))
    for ctx in syms:
        row = [F.get(ctx, {}).get(s, 0) for s in syms]
        i = 0
        while i < len(row):
            table += uint7(row[i])
            if row[i] == 0:
                run = 0
                while i + 1 + run < len(row) and row[i + 1 + run] == 0 and run < 255:
                    run += 1
                table.append(run)
                i += run
            i += 1
    if code_table:
        coded = order0(bytes(table), 4)
        head = bytes([(bits << 4) | 1]) + uint7(len(table)) + uint7(len(coded)) + coded
    else:
        head = bytes([bits << 4]) + bytes(table)
    return head + _emit([(j, F[ctx][b], C[ctx][b]) for j, ctx, b in walk], n_states, bits)


def pack(data):
    """-> (symbol map, packed bytes) or None when the data has more than 16 distinct symbols"""
    syms = sorted(set(data))
    if len(syms) > 16:
        return None
    if len(syms) <= 1:
        return bytes(syms), b""
    width = 1 if len(syms) <= 2 else 2 if len(syms) <= 4 else 4
    per, code = 8 // width, {s: i for i, s in enumerate(syms)}
    out = bytearray((len(data) + per - 1) // per)
    for i, b in enumerate(data):
        out[i // per] |= code[b] << ((i % per) * width)  # low bits first
    return bytes(syms), bytes(out)


def rle(data):
    """-> (side stream, literals): symbols that ever repeat are on the list; each of their runs is one literal + (length - 1)"""
    if not data:
        return b"", b""
    listed = {data[i] for i in range(1, len(data)) if data[i] == data[i - 1]} or {data[0]}
    meta = bytearray([len(listed) & 0xFF]) + bytes(sorted(listed))
    lits, i = bytearray(), 0
    while i < len(data):
        b = data[i]
        lits.append(b)
        if b in listed:
            run = 1
            while i + run < len(data) and data[i + run] == b:
                run += 1
            meta += uint7(run - 1)
            i += run
        else:
            i += 1
    return bytes(meta), bytes(lits)


def encode(data, flags=0, stripes=4, o1_bits=12, code_table=False, code_rle_meta=True):
    """one rANS Nx16 stream of `data` with the transforms `flags` names (PACK is dropped when the data has > 16 symbols)"""
    data = bytes(data)
    if flags & PACK and pack(data) is None:
        flags &= ~PACK
    if flags & STRIPE:
        flags &= STRIPE | NOSZ | X32 | ORDER  # the transforms apply inside the sub-streams
    n_states = 32 if flags & X32 else 4
    out = bytearray([flags])
    if not flags & NOSZ:
        out += uint7(len(data))
    if flags & STRIPE:
        inner = (flags & (X32 | ORDER)) | NOSZ
        subs = [encode(data[j::stripes], inner, o1_bits=o1_bits) for j in range(stripes)]
        out.append(stripes)
        for s in subs:
            out += uint7(len(s))
        return bytes(out) + b"".join(subs)
    cur = data
    if flags & PACK:
        pmap, cur = pack(cur)
        out += bytes([len(pmap)]) + pmap + uint7(len(cur))
    if flags & RLE:
        meta, cur = rle(cur)
        if code_rle_meta and meta:
            coded = order0(meta, n_states)
            out += uint7(2 * len(meta)) + uint7(len(cur)) + uint7(len(coded)) + coded
        else:
            out += uint7(2 * len(meta) + 1) + uint7(len(cur)) + meta
    if flags & CAT:
        out += cur
    elif flags & ORDER:
        out += order1(cur, n_states, o1_bits, code_table)
    else:
        out += order0(cur, n_states)
    return bytes(out)
