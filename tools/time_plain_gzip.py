#!/usr/bin/env python3
"""End-to-end time of a PLAIN gzip (one member, not BGZF) FASTQ / VCF through the GPU pipeline, next to its BGZF twin and the
host-zlib path (EXON_HIP_GPU_GZIP=0).  The one-member file is written pigz-style: pieces deflated on all cores with a full flush
between them, one header, one trailer (CRC-32 / ISIZE over the whole text).
usage: time_plain_gzip.py {fastq|vcf} ROWS [runs] [--host]   (EXON_HIP_PIPE_TRACE=1: per-phase split on stderr)"""
import os
import struct
import subprocess
import sys
import time
import zlib
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PIECE = 64 << 20


def _deflate_piece(args):
    path, off, n, last, level = args
    with open(path, "rb") as f:
        f.seek(off)
        data = f.read(n)
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    out = co.compress(data) + (co.flush(zlib.Z_FINISH) if last else co.flush(zlib.Z_FULL_FLUSH))
    return out, zlib.crc32(data), len(data)


def crc32_combine(c1, c2, len2):
    """zlib's crc32_combine (not exposed by Python's zlib): GF(2) matrix squaring"""
    def times(mat, vec):
        s, i = 0, 0
        while vec:
            if vec & 1:
                s ^= mat[i]
            vec >>= 1
            i += 1
        return s

    def square(mat):
        return [times(mat, mat[n]) for n in range(32)]
    if len2 <= 0:
        return c1
    odd = [0xEDB88320] + [1 << n for n in range(31)]
    even = square(odd)
    odd = square(even)
    while True:
        even = square(odd)
        if len2 & 1:
            c1 = times(even, c1)
        len2 >>= 1
        if not len2:
            break
        odd = square(even)
        if len2 & 1:
            c1 = times(odd, c1)
        len2 >>= 1
        if not len2:
            break
    return c1 ^ c2


def write_plain_gzip(src, dst, level=6, workers=None):
    size = os.path.getsize(src)
    jobs = [(src, o, min(PIECE, size - o), o + PIECE >= size, level) for o in range(0, size, PIECE)]
    crc, total = 0, 0
    with open(dst, "wb") as f, ProcessPoolExecutor(workers or min(16, os.cpu_count() or 4)) as ex:
        f.write(b"\x1f\x8b\x08\x00\0\0\0\0\x00\xff")
        for out, c, n in ex.map(_deflate_piece, jobs):
            f.write(out)
            crc = crc32_combine(crc, c, n) if total else c
            total += n
        f.write(struct.pack("<II", crc & 0xFFFFFFFF, total & 0xFFFFFFFF))


def main():
    import exon_amd
    kind, rows = sys.argv[1], int(float(sys.argv[2]))
    runs = int(sys.argv[3]) if len(sys.argv) > 3 and not sys.argv[3].startswith("-") else 5
    d = os.environ.get("EXON_TMP", "/dev/shm")
    text = os.path.join(d, f"pg_{kind}_{rows}.{kind}")
    gz, bgz = text + ".plain.gz", text + ".bgzf.gz"
    t0 = time.perf_counter()
    if not os.path.exists(text):
        subprocess.check_call([os.path.join(ROOT, "tools", "bin", "gen_text"), kind, str(rows), text] + (["100", "0"] if kind == "fastq" else []))
    if not os.path.exists(gz):
        write_plain_gzip(text, gz)
    if not os.path.exists(bgz):
        subprocess.check_call([os.path.join(ROOT, "tools", "bin", "bgzip"), text, bgz, "6"])
    print(f"files ready in {time.perf_counter() - t0:.1f} s: text {os.path.getsize(text) / 1e9:.2f} GB, plain gzip {os.path.getsize(gz) / 1e9:.2f} GB, BGZF {os.path.getsize(bgz) / 1e9:.2f} GB", flush=True)
    ctx = exon_amd.Context(0)

    def once(path, gpu_parse=True):
        t0 = time.perf_counter()
        if kind == "vcf":
            scan = exon_amd.Scan(path, "vcf", info_field="AF", gpu_parse=gpu_parse)
            plan = ctx.plan_cmp_avg_by_group(">", 0.01, 64, columns=(4, 2, 3))
        else:
            scan = exon_amd.Scan(path, "fastq", gpu_parse=gpu_parse)
            plan = ctx.plan_qual_pos_hist(256, columns=(3,))
        st = plan.open()
        t1 = time.perf_counter()
        n = st.consume(scan)
        t2 = time.perf_counter()
        counts, sums = st.finish()
        flags = scan.decoded_on_gpu()
        st.close()
        plan.close()
        scan.close()
        return n, t2 - t1, time.perf_counter() - t0, flags, int(sum(int(c) for c in counts))

    def report(name, path, n_runs, **kw):
        res = [once(path, **kw) for _ in range(n_runs)]
        warm = res[1:] if len(res) > 1 else res
        cs = sorted(r[1] for r in warm)
        print(f"{name}: {res[0][0]} rows, sum(counts) {res[0][4]}, decoded/inflated on the GPU {tuple(res[0][3])}: consume best {cs[0] * 1e3:.1f} ms, median "
              f"{cs[len(cs) // 2] * 1e3:.1f} ms, first {res[0][1] * 1e3:.1f} ms; open..close best {min(r[2] for r in warm) * 1e3:.1f} ms", flush=True)
        return res[0][4]
    a = report("plain gzip, GPU inflate", gz, runs)
    b = report("BGZF twin, GPU inflate ", bgz, runs)
    assert a == b, (a, b)
    if "--host" in sys.argv:
        os.environ["EXON_HIP_GPU_GZIP"] = "0"
        c = report("plain gzip, host zlib   ", gz, 2)
        assert a == c
    if "--keep" not in sys.argv:
        for p in (text, gz, bgz):
            os.remove(p)


if __name__ == "__main__":
    main()
