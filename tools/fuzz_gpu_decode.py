#!/usr/bin/env python3
"""Corrupted files through the GPU decode pipelines (inflate + record splitting + parse on the device): every input must
either give the host decoder's answer or raise -- never hang, never a different answer.  usage: fuzz_gpu_decode.py [n_per_format]"""
import os, random, struct, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import exon_amd
from bgzf_index_writer import bgzf_blocks
FX = os.path.join(ROOT, "tests", "golden", "ref_fixtures")


def bgzf_file(data, chunk=65280):
    out = []
    for i in list(range(0, len(data), chunk)) + [None]:
        d = b"" if i is None else data[i:i + chunk]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        c = co.compress(d) + co.flush()
        bs = 12 + 6 + len(c) + 8
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff" + struct.pack("<H", 6) + b"BC\x02\0" + struct.pack("<H", bs - 1) + c +
                   struct.pack("<II", zlib.crc32(d) & 0xFFFFFFFF, len(d)))
    return b"".join(out)


def answer(ctx, path, fmt, gpu):
    if fmt in ("vcfinfo", "bcfinfo"):  # typed INFO keys + the config-4 shape: WHERE info.DP > 1, AVG(qual), COUNT(*) GROUP BY filter
        scan = exon_amd.Scan(path, fmt[:3], info_field="DP,MQ0F,MQSB", gpu_parse=gpu)
        plan = ctx.plan_cmp_avg_by_group(">", 1.0, 64, columns=(4, 2, 3))
        st = plan.open()
        try:
            rows = st.consume(scan)
            counts, sums = st.finish()
            # FILTER ids are interned in first-seen order, which may differ between the paths: compare the multiset
            groups = sorted((int(counts[g]), int(counts[64 + g]), round(float(sums[g]), 3)) for g in range(64) if counts[64 + g])
            return rows, tuple(groups)
        finally:
            st.close(); plan.close(); scan.close()
    if fmt in ("vcfreg", "bamreg"):  # a pushed-down region without an index: the whole file is decoded, rows are masked
        scan = exon_amd.Scan(path, fmt[:3], region="1:10000000-10000100" if fmt == "vcfreg" else "chr1:1-12209145", use_index=False, gpu_parse=gpu)
        fmt = fmt[:3]
    else:
        scan = exon_amd.Scan(path, fmt, gpu_parse=gpu)
    if fmt == "fastq":
        plan = ctx.plan_qual_pos_hist(256, columns=(3,))
    elif fmt in ("bam", "sam"):
        plan = ctx.plan_flag_mapq_group_count(0, 0, 0, max(1, scan.dictionary_size(2)), columns=(0, 1, 2))
    else:
        plan = ctx.plan_region_count(0, 1, None, columns=(0, 1))
    st = plan.open()
    try:
        rows = st.consume(scan)
        counts, _ = st.finish()
        return rows, tuple(int(x) for x in counts)
    finally:
        st.close(); plan.close(); scan.close()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rnd = random.Random(int(os.environ.get("FUZZ_SEED", 9)))
    ctx = exon_amd.Context(0)
    only = os.environ.get("FUZZ_ONLY")
    srcs = {"vcfreg": ("vcf/index.vcf.gz", True), "bamreg": ("bam/test.bam", True), "vcfinfo": ("vcf/index.vcf.gz", True), "bcfinfo": ("bcf/index.bcf", True), "bam": ("bam/test.bam", True), "bcf": ("bcf/index.bcf", True), "vcf": ("vcf/index.vcf.gz", True), "sam": ("sam/test.sam", False)}
    for fmt, (rel, framed) in srcs.items():
        good = open(os.path.join(FX, rel), "rb").read()
        raw = b"".join(d for _, _, d in bgzf_blocks(good)) if framed else good
        same = both_err = gpu_only_err = 0
        for it in range(n):
            b = bytearray(raw)
            lo = 0 if rnd.random() < 0.2 else min(len(b) - 1, 1500)
            for _ in range(rnd.choice([1, 1, 2, 4, 16])):
                b[rnd.randrange(lo, len(b))] = rnd.randrange(256)
            if rnd.random() < 0.1:
                b = b[:rnd.randrange(1, len(b))]
            if only and fmt != only:  # same random stream, nothing run
                continue
            path = f"/tmp/fz_gpu.{fmt[:3]}" + (".gz" if fmt.startswith("vcf") else "")
            open(path, "wb").write(bgzf_file(bytes(b)) if framed else bytes(b))
            if os.environ.get("FUZZ_KEEP"):  # the input that is running when the process dies
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                open(os.path.join(ROOT, "gpurun_out", "fuzz_last." + fmt), "wb").write(bytes(b))
                print(fmt, it, flush=True)
            if os.environ.get("FUZZ_SAVE_DIR"):
                open(os.path.join(os.environ["FUZZ_SAVE_DIR"], f"{fmt}_{it:03d}"), "wb").write(bytes(b))
            try:
                h = answer(ctx, path, fmt, False)
            except exon_amd.ExonHipError:
                h = None
            try:
                g = answer(ctx, path, fmt, True)
            except exon_amd.ExonHipError:
                g = None
            if h is None and g is None:
                both_err += 1
            elif g is None:
                gpu_only_err += 1
            else:
                assert g == h, (fmt, it, g, h)
                same += 1
        print(fmt, "same answer", same, "both rejected", both_err, "only the GPU path rejected", gpu_only_err, flush=True)


def big():
    """multi-slab inputs: synthetic files of 150 k records, 1 MB slabs, corrupted before framing"""
    import subprocess
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    os.environ["EXON_HIP_GPU_PARSE_SLAB_MB"] = "1"
    rnd = random.Random(int(os.environ.get("FUZZ_SEED", 21)))
    ctx = exon_amd.Context(0)
    gen = os.path.join(ROOT, "tools", "bin", "gen_text")
    for fmt, kind, framed in (("vcf", "vcf", True), ("bam", "bam", True), ("sam", "sam", False), ("fastq", "fastq", False), ("fastq", "fastq", True), ("bcf", "bcf", True)):
        subprocess.check_call([gen, kind, "150000", "/tmp/fz_big.raw"] + (["100"] if kind in ("bam", "sam") else []))
        raw = open("/tmp/fz_big.raw", "rb").read()
        same = both_err = gpu_only_err = 0
        for it in range(n):
            b = bytearray(raw)
            for _ in range(rnd.choice([0, 1, 2, 8, 64])):
                b[rnd.randrange(min(len(b) - 1, 3000), len(b))] = rnd.randrange(256)
            if rnd.random() < 0.15:
                b = b[:rnd.randrange(len(b) // 2, len(b))]
            ext = {"vcf": ".vcf.gz", "bam": ".bam", "bcf": ".bcf", "sam": ".sam", "fastq": ".fastq.gz" if framed else ".fastq"}[fmt]
            path = "/tmp/fz_big" + ext
            open(path, "wb").write(bgzf_file(bytes(b)) if framed else bytes(b))
            try:
                h = answer(ctx, path, fmt, False)
            except exon_amd.ExonHipError:
                h = None
            try:
                g = answer(ctx, path, fmt, True)
            except exon_amd.ExonHipError:
                g = None
            if h is None and g is None:
                both_err += 1
            elif g is None:
                gpu_only_err += 1
            else:
                assert g == h, (fmt, it, g and g[0], h and h[0])
                same += 1
        print("big", fmt, "framed" if framed else "plain", "same answer", same, "both rejected", both_err, "only the GPU path rejected", gpu_only_err, flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        big()
    else:
        main()
