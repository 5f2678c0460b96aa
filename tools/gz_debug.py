"""Debug aid for gzip_stream.hip: decode with and without CRC verification, report the first difference from zlib."""
import gzip, os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import exon_amd

def check(ctx, name, raw, **kw):
    want = gzip.decompress(raw)
    for verify in ("0", "1"):
        os.environ["EXON_HIP_GZ_VERIFY_CRC"] = verify
        try:
            got, st = ctx.gzip_inflate(raw, return_stats=True, **kw)
        except exon_amd.ExonHipError as e:
            print(f"{name} verify={verify}: ERROR {e}")
            continue
        if got == want:
            print(f"{name} verify={verify}: ok {len(want)} bytes {st}")
        else:
            n = min(len(got), len(want))
            d = next((i for i in range(n) if got[i] != want[i]), n)
            print(f"{name} verify={verify}: MISMATCH len got {len(got)} want {len(want)} first diff at {d}: got {got[d:d+16]!r} want {want[d:d+16]!r} {st}")

if __name__ == "__main__":
    ctx = exon_amd.Context(0)
    fx = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_fixtures")
    for n in ("fastq/test.fastq.gz", "fasta/test.fasta.gz"):
        check(ctx, n, open(os.path.join(fx, n), "rb").read())
    rng = np.random.default_rng(1)
    text = b"".join(b"chr%d\t%d\tAF=%f\n" % (i % 22, i * 37, rng.random()) for i in range(50000))
    def gz(data, level=6, strategy=0):
        co = zlib.compressobj(level, zlib.DEFLATED, 31, 8, strategy)
        return co.compress(data) + co.flush()
    check(ctx, "tiny", gz(b"ACGT\n"))
    check(ctx, "text-1chunk", gz(text[:20000]))
    check(ctx, "text", gz(text))
    check(ctx, "text-stored", gz(text, 0))
    check(ctx, "text-fixed", gz(text, 6, zlib.Z_FIXED))
    check(ctx, "text-slabs", gz(text), slab_bytes=128 << 10, out_cap=8 << 20)
    os.environ["EXON_HIP_GZ_CHUNK_KB"] = "4"
    check(ctx, "text-4k", gz(text))
    check(ctx, "text-4k-slabs", gz(text), slab_bytes=48 << 10, out_cap=8 << 20)
    check(ctx, "random-4k", gz(rng.integers(0, 256, 300000, dtype=np.uint8).tobytes()))
    check(ctx, "zeros-4k", gz(bytes(300000)))
