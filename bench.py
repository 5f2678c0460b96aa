#!/usr/bin/env python3
"""bench.py -- headline benchmark: filter + GROUP BY aggregate over a synthetic VCF (BASELINE.json config 4).

    SET exon.vcf_parse_info = true;
    SELECT filter, AVG(qual), COUNT(*) FROM v WHERE info."AF" > 0.01 GROUP BY filter

One "step" = one full pass of the fused filter+aggregate kernel over this rank's HBM-resident shard (ONE launch: the last
workgroup folds the per-workgroup records and WRITES the state -- EXON_HIP_LAUNCH_OVERWRITE, no zeroing pass) followed --
when N > 1 -- by the merge of the packed partial states (5 x {2 x i64 count, f64 sum} = 120 B): ONE RCCL all-gather and a
fold in rank order.

Scaling ("1 B rows, 8 equal shards"):
  --scaling strong (default)  --rows is the TOTAL (default 1e9); rank k owns rows [k N/W, (k+1) N/W)
  --scaling weak              --rows is per GPU; the table grows with the number of GPUs
Launch: `python bench.py --gpus N` (N > 1 without a launcher: the script starts its own N ranks) or
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N --steps K --warmup W`.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (imported before the HIP library so both share one HIP runtime)
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
C5_L = int(os.environ.get("EXON_BENCH_C5_L", 100))  # config 5's read length (BASELINE.json: 100)
BYTES_PER_ROW = {"c4": 12.25, "c2": 12.0, "c3": 9.25, "c5": 4.0 + C5_L, "c6": 20.375}  # algorithmic bytes/row, SURVEY.md section 8(d)
SEED = {"c2": 2, "c3": 3, "c4": 4, "c5": 5, "c6": 6}
DTYPE = {"c4": "f64", "c2": "int64", "c3": "int64", "c5": "u8", "c6": "int64"}
WORKLOAD = {"c4": "config 4: 1B-row synthetic VCF, WHERE info.AF > 0.01, AVG(qual), COUNT(*) GROUP BY filter",
            "c2": "config 2: synthetic VCF, chrom='7' AND pos in [5e7,1e8], COUNT(*)",
            "c3": "config 3: synthetic BAM, flag&1284=0 AND mapq>=30, COUNT(*) GROUP BY reference",
            "c5": f"config 5: synthetic FASTQ (L={C5_L}), per-position 256-bin quality histogram; rows = reads",
            "c6": "synthetic alignments, bam_region_filter('<ref 7>:50000000-100000000', reference, start, end), COUNT(*)"}
ARITH = {"c4": "f32 columns compared as totalOrder i32 keys, f64 sums, i64 counts",
         "c2": "i32 / i64 compares, i64 count", "c3": "i32 mask compare, u8 compare, i64 counts",
         "c5": "u8 bytes, u32 LDS counters folded into i64", "c6": "i32 / i64 compares, i64 count"}
GENERATOR_NOTE = {
    "c4": "counter-based generator (DESIGN.md section 5); deviates from SURVEY 8(d): AF is log-uniform BY OCTAVE over "
          "[2^-14, 1) with 0.01f planted at p=1/1024 instead of 10^U[-4,0] (bit-identical on CPU and GPU without libm); "
          "47.5 % of the rows pass AF > 0.01 instead of 50 %; NULL rates and FILTER mix as in SURVEY"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the chip reaches its steady clocks some tens of milliseconds into a run (same box, alternating: 3 warm-up steps +
    # 20 timed 1.888-1.891 ms per step, 20 + 20 1.864-1.871, 50 + 100 1.841-1.873: profiles/r5_bench_warmup_sensitivity.log)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--rows", type=float, default=1e9, help="total rows (strong scaling) / rows per GPU (weak scaling)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--merge", default="auto", choices=["auto", "native", "torch"],
                    help="N > 1: native = ncclAllGather on the kernels' stream through the C ABI; torch = torch.distributed; "
                         "auto = native when it initialises and reproduces torch's result during warm-up, else torch")
    ap.add_argument("--workload", default="c4", choices=["c4", "c2", "c3", "c5", "c6"])
    ap.add_argument("--groups", type=int, default=5,
                    help="c4 only: number of GROUP BY keys.  5 = config 4's FILTER mix (ids in registers); > 8 keeps ids 0..3 in "
                         "registers, ids 4..4099 in the LDS table and partitions ids >= 4100 by id range -- the paths files and "
                         "high-cardinality keys take")
    ap.add_argument("--group-dist", default="zipf", choices=["zipf", "uniform"],
                    help="--groups != 5: key frequencies.  zipf = log-uniform ids (P(id < k) = ln(k+1)/ln(G+1): dictionary ids "
                         "are handed out in order of first appearance, so frequent keys are early); uniform = the worst case")
    ap.add_argument("--cpu-sample-rows", type=float, default=128e6)
    ap.add_argument("--cpu-reps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic in this run (two rocprofv3 --pmc passes of a short child run); use profiles/traffic.json")
    ap.add_argument("--no-extras", action="store_true", help="skip the side measurements (stated-size configs, H2D-inclusive rate)")
    ap.add_argument("--probe-reps", type=int, default=20, help="passes of the bare read probe before and after the timed steps")
    ap.add_argument("--calib-copy", action="store_true",
                    help="(child runs of measure_traffic) one 1 GiB device-to-device copy first: the counter pass calibrates its bytes-per-count on it")
    return ap.parse_args()


class Workload:
    """Device-resident synthetic shard + one launch of the hot path on torch's current stream.

    The partial state is ONE packed int64-typed tensor `[n_i64 counters][n_f64 sums bit-cast]` (exon_hip_plan_state_size)
    written by exon_hip_plan_launch; `counts` / `sums` are views of it."""

    def __init__(self, ctx, kind, rows, row0, n_total, groups=5, group_dist="zipf"):
        self.ctx, self.kind, self.n = ctx, kind, rows
        dev = torch.device("cuda", torch.cuda.current_device())
        s = torch.cuda.current_stream().cuda_stream
        lib, h = ctx.lib, ctx.h
        nb = (rows + 7) // 8 + 64
        alloc = max(rows, 16)
        if kind == "c4":
            self.af = torch.empty(alloc, dtype=torch.float32, device=dev)
            self.qual = torch.empty(alloc, dtype=torch.float32, device=dev)
            self.fid = torch.empty(alloc, dtype=torch.int32, device=dev)
            self.av = torch.zeros(nb, dtype=torch.uint8, device=dev)
            self.qv = torch.zeros(nb, dtype=torch.uint8, device=dev)
            ctx._check(lib.exon_hip_gen_c4(h, s, SEED["c4"], row0, row0 + rows, self.af.data_ptr(), self.av.data_ptr(),
                                           self.qual.data_ptr(), self.qv.data_ptr(), self.fid.data_ptr()))
            self.G = groups
            if groups != 5:
                synth_group_ids(self.fid, rows, row0, groups, group_dist)
            self.plan = ctx.plan_cmp_avg_by_group(">", 0.01, self.G)
            self.cols = [(self.af.data_ptr(), self.av.data_ptr(), None), (self.qual.data_ptr(), self.qv.data_ptr(), None),
                         (self.fid.data_ptr(), None, None)]
        elif kind == "c2":
            self.chrom = torch.empty(alloc, dtype=torch.int32, device=dev)
            self.pos = torch.empty(alloc, dtype=torch.int64, device=dev)
            ctx._check(lib.exon_hip_gen_c2(h, s, SEED["c2"], n_total, row0, row0 + rows, self.chrom.data_ptr(),
                                           self.pos.data_ptr()))
            self.plan = ctx.plan_region_count(6, 50000000, 100000000)
            self.cols = [(self.chrom.data_ptr(), None, None), (self.pos.data_ptr(), None, None)]
        elif kind == "c6":
            self.ref = torch.empty(alloc, dtype=torch.int32, device=dev)
            self.start = torch.empty(alloc, dtype=torch.int64, device=dev)
            self.end = torch.empty(alloc, dtype=torch.int64, device=dev)
            self.rv = torch.zeros(nb, dtype=torch.uint8, device=dev)
            self.pv = torch.zeros(nb, dtype=torch.uint8, device=dev)
            ctx._check(lib.exon_hip_gen_c6(h, s, SEED["c6"], row0, row0 + rows, self.ref.data_ptr(), self.rv.data_ptr(),
                                           self.start.data_ptr(), self.end.data_ptr(), self.pv.data_ptr()))
            self.plan = ctx.plan_overlap_count(6, 50000000, 100000000, columns=(0, 1, 2))
            self.cols = [(self.ref.data_ptr(), self.rv.data_ptr(), None), (self.start.data_ptr(), self.pv.data_ptr(), None),
                         (self.end.data_ptr(), self.pv.data_ptr(), None)]
        elif kind == "c3":
            self.flag = torch.empty(alloc, dtype=torch.int32, device=dev)
            self.mapq = torch.empty(alloc + 64, dtype=torch.uint8, device=dev)
            self.ref = torch.empty(alloc, dtype=torch.int32, device=dev)
            self.mv = torch.zeros(nb, dtype=torch.uint8, device=dev)
            self.rv = torch.zeros(nb, dtype=torch.uint8, device=dev)
            ctx._check(lib.exon_hip_gen_c3(h, s, SEED["c3"], row0, row0 + rows, self.flag.data_ptr(), self.mapq.data_ptr(),
                                           self.mv.data_ptr(), self.ref.data_ptr(), self.rv.data_ptr()))
            self.R = 25
            self.plan = ctx.plan_flag_mapq_group_count(1284, 0, 30, self.R)
            self.cols = [(self.flag.data_ptr(), None, None), (self.mapq.data_ptr(), self.mv.data_ptr(), None),
                         (self.ref.data_ptr(), self.rv.data_ptr(), None)]
        elif kind == "c5":
            # FASTQ quality strings, L = 100.  Arrow Utf8 has int32 offsets, so the shard is a sequence of batches, each as
            # large as such a column gets (21,474,835 reads at L = 100; EXON_BENCH_C5_BATCH overrides), packed back to back in
            # one allocation -- so most batches start at an odd address.  (Round 2 used 20 M-read batches because full ones ran
            # 25 % slower; the cause was that their first bytes were not 256-byte aligned and the kernel's 256-byte rows then
            # straddled cache lines.  Round 3's kernel reads memory-aligned rows whatever the base.)  The histogram accumulates.
            self.L = C5_L  # 100 = BASELINE.json's config; EXON_BENCH_C5_L times other uniform read lengths (151, 250 ...)
            self.batch = max(1, min(rows, int(os.environ.get("EXON_BENCH_C5_BATCH", (2**31 - 64) // self.L))))
            self.bytes = torch.empty(rows * self.L + 64, dtype=torch.uint8, device=dev)
            nb = (rows + self.batch - 1) // self.batch
            self.off = torch.empty(rows + nb + 4, dtype=torch.int32, device=dev)  # every batch has its own offsets buffer
            self.chunks = []
            for k, b0 in enumerate(range(0, rows, self.batch)):
                nb_ = min(self.batch, rows - b0)
                d_off = self.off.data_ptr() + 4 * (b0 + k)
                d_bytes = self.bytes.data_ptr() + b0 * self.L
                ctx._check(lib.exon_hip_gen_c5(h, s, SEED["c5"], row0 + b0, row0 + b0 + nb_, self.L, d_off, d_bytes))
                self.chunks.append(([(d_bytes, None, d_off)], nb_))
            self.fused = os.environ.get("EXON_BENCH_C5_FUSED", "1") != "0"
            self.plan = ctx.plan_qual_pos_hist(self.L)
        self.n_i64, self.n_f64 = self.plan.n_i64, self.plan.n_f64
        self._go = None
        self.state = torch.zeros(self.n_i64 + self.n_f64, dtype=torch.int64, device=dev)
        self.counts = self.state[:self.n_i64]
        self.sums = self.state[self.n_i64:].view(torch.float64) if self.n_f64 else None
        torch.cuda.synchronize()

    def probe_buffers(self):
        """(device pointers, bytes of each) of up to 4 value buffers of this shard for exon_hip_read_probe: the plan's columns, all
        cut to the shortest one's length in bytes (the probe walks its buffers in lock-step)."""
        t = {"c4": [self.af, self.qual, self.fid] if self.kind == "c4" else None,
             "c2": [self.chrom, self.pos] if self.kind == "c2" else None,
             "c6": [self.ref, self.start, self.end] if self.kind == "c6" else None,
             "c3": [self.flag, self.ref] if self.kind == "c3" else None,
             "c5": [self.bytes] if self.kind == "c5" else None}[self.kind]
        nbytes = min(x.numel() * x.element_size() for x in t)
        if nbytes < (64 << 10):
            return None, 0
        return [x.data_ptr() for x in t], nbytes

    def run(self):
        """The hot path over the shard: main + finalize kernels; the state is DEFINED by the launch (overwrite mode),
        so there is no zeroing pass.  c5: the shard is a list of full Arrow batches handed over in one call."""
        s = torch.cuda.current_stream().cuda_stream
        if self.kind == "c5":
            if self.fused:  # exon_hip_plan_launch_chunks: one offsets scan + one main kernel + one fold per 64 batches
                self.plan.launch_chunks(self.chunks, self.state.data_ptr(), overwrite=True, stream=s)
            else:           # one launch (scan + main + fold) per batch
                for k, (cols, nb_) in enumerate(self.chunks):
                    self.plan.launch(cols, nb_, self.state.data_ptr(), overwrite=(k == 0), stream=s)
        else:
            if self._go is None or self._go_stream != s:  # arguments marshalled once per (table, stream)
                self._go = self.plan.prepared(self.cols, self.n, self.state.data_ptr(), overwrite=True, stream=s)
                self._go_stream = s
            self._go()


def synth_group_ids(fid, rows, row0, groups, dist):
    """--groups G: overwrite the FILTER ids with G synthetic keys (seeded per 16 Mi-row chunk by its global start row, so a
    shard of a multi-GPU run holds the same ids as the same rows of a 1-GPU run when shards start on chunk boundaries)."""
    import math
    step = 1 << 24
    for c0 in range(0, rows, step):
        m = min(step, rows - c0)
        g = torch.Generator(device=fid.device)
        g.manual_seed(0x5EED0000 + (row0 + c0) // step)
        u = torch.rand(m, generator=g, device=fid.device, dtype=torch.float32)
        if dist == "uniform":
            ids = (u * groups).to(torch.int32)
        else:
            ids = (torch.exp(u.double() * math.log(groups + 1.0)) - 1.0).to(torch.int32)
        fid[c0:c0 + m] = ids.clamp_(0, groups - 1)


def torch_reference_c4(wl, thr=0.01):
    """Plain torch fp64 statement of config 4's query over a Workload's device columns (used when --groups != 5, where the
    C oracle has no generator): (count(y)[G], count(*)[G], sum(y)[G])."""
    n, G = wl.n, wl.G
    idx = torch.arange(n, device=wl.af.device)
    av = ((wl.av[idx >> 3] >> (idx & 7).to(torch.uint8)) & 1).bool()
    qv = ((wl.qv[idx >> 3] >> (idx & 7).to(torch.uint8)) & 1).bool()
    keep = av & (wl.af[:n].double() > thr)
    gid = wl.fid[:n].long()
    crow = torch.bincount(gid[keep], minlength=G)
    kq = keep & qv
    cnn = torch.bincount(gid[kq], minlength=G)
    sums = torch.zeros(G, dtype=torch.float64, device=gid.device).index_add_(0, gid[kq], wl.qual[:n][kq].double())
    return cnn, crow, sums


def host_cpus():
    """(usable, logical, quota): CPUs this process can keep busy = logical CPUs of its affinity mask capped by the container's
    CFS quota (cgroup v2 cpu.max / v1 cfs_quota_us).  The GPU boxes of this pool show 256 logical CPUs under a quota of 16:
    128 threads there ran 12.6 cores' worth of work, 16 threads run 15.6 (tools/host_scaling.py, profiles/r4_host_scaling.log)."""
    logical = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    usable = logical if quota is None else max(1, min(logical, int(quota + 0.999)))
    return usable, logical, quota


def cpu_baseline(kind, sample_rows, n_total, reps):
    """The CPU restatement of the Exon/DataFusion plan (oracle/exon_oracle.c) on a bounded sample of the
    SAME synthetic rows [0, sample_rows), all host cores; returns (result dict, oracle outputs)."""
    from oracle import Oracle
    orc = Oracle()
    n = int(sample_rows)
    secs, mat = [], []
    T, logical, quota = host_cpus()   # partitions = the CPUs the container may really use (target_partitions = num_cpus)
    if kind == "c4":
        af, av, q, qv, fid = orc.gen_c4(SEED["c4"], 0, n)
        for _ in range(reps):
            s, cn, cr, t = orc.c4_cmp_avg_by_group(af, av, q, qv, fid, orc.c4_filters(), 0.01, ">", threads=T)
            secs.append(t.seconds_exec), mat.append(t.seconds_materialize)
        out = (np.concatenate([cn, cr]), s)
    elif kind == "c2":
        c, p = orc.gen_c2(SEED["c2"], n_total, 0, n)
        for _ in range(reps):
            r, t = orc.c2_region_count(c, p, orc.c2_contigs(), "7:50000000-100000000", threads=T)
            secs.append(t.seconds_exec), mat.append(t.seconds_materialize)
        out = (np.array([r], np.int64), None)
    elif kind == "c6":
        import time as _t

        class _T:  # the column-wise numpy restatement is single-threaded
            threads = 1
        n = min(n, 32_000_000)
        ref, rv, st, en, pv = orc.gen_c6(SEED["c6"], 0, n)
        t = _T()
        names = [orc.c3_refs()[i] for i in range(25)]
        for _ in range(reps):
            t0 = _t.perf_counter()
            r = orc.c6_overlap_count(ref, rv, st, pv, en, pv, names, names[6] + ":50000000-100000000")
            secs.append(_t.perf_counter() - t0), mat.append(0.0)
        out = (np.array([r], np.int64), None)
    elif kind == "c5":
        n = min(n, 8_000_000)
        off, data = orc.gen_c5(SEED["c5"], 0, n, C5_L)
        for _ in range(reps):
            hcpu, t = orc.c5_qual_pos_hist(off, data, C5_L, threads=T)
            secs.append(t.seconds_exec), mat.append(t.seconds_materialize)
        out = (hcpu.reshape(-1), None)
    else:
        f, mq, mv, ref, rv = orc.gen_c3(SEED["c3"], 0, n)
        for _ in range(reps):
            cnt, t = orc.c3_flag_mapq_group_count(f, mq, mv, ref, rv, orc.c3_refs(), 1284, 0, 30, threads=T)
            secs.append(t.seconds_exec), mat.append(t.seconds_materialize)
        out = (cnt, None)
    best = float(np.median(secs))
    # single-thread figure on a slice of the same rows (SURVEY section 8d asks for both)
    one = None
    if kind == "c4":
        m = min(n, 8_000_000)
        _, _, _, t1 = orc.c4_cmp_avg_by_group(af[:m], av[:(m + 7) // 8], q[:m], qv[:(m + 7) // 8], fid[:m], orc.c4_filters(),
                                              0.01, ">", threads=1)
        one = round(m / t1.seconds_exec / 1e6, 2)
    res = {"value": round(n / best / 1e6, 2), "unit": "Mrows/s", "cores": t.threads, "kind": "port",
           # the stable figure of this leg: one partition on one core (box to box within a few percent); `value` = that x the
           # cores the container may use x the parallel efficiency below
           "single_thread_value": one,
           "parallel_speedup": round(n / best / 1e6 / one, 1) if one else None,
           "core_seconds": round(best * t.threads, 3),
           "host": {"logical_cpus": logical, "cfs_quota_cpus": quota, "partitions": t.threads},
           "sample": f"rows [0,{n}) of the same synthetic table as 8192-row Arrow-layout batches (Utf8/List<Utf8> keys), "
                     f"{t.threads} partitions = the CPUs this container may use ({logical} logical CPUs"
                     + (f", CFS quota {quota:g}: threads beyond the quota are throttled every 100 ms period -- 128 threads "
                        f"delivered 12.6 cores' worth in round 3" if quota else ", no CFS quota")
                     + f"); median of {reps} runs, {best:.3f}s exec each "
                     f"(+{float(np.median(mat)):.2f}s untimed Arrow-layout build); {best * t.threads:.2f} core-seconds per run"}
    return res, out


PMC_KERNEL = {"c4": "k4_cmp_avg_by_group_main", "c2": "k2_region_count_main", "c3": "k3_flag_mapq_group_count_main",
              "c5": "k5_main", "c6": "k6_overlap_count_main"}


def measure_traffic(a, rows):
    """roofline.traffic measured in THIS run: HBM bytes per launch of the dominant kernel from rocprofv3 PMC counters, collected
    the way /opt/skills/guides/MI355X_MICROARCH.md prescribes -- FETCH_SIZE and WRITE_SIZE in SEPARATE passes, each with
    --kernel-trace only, values in KiB.  On gfx950 FETCH_SIZE counts about half of the bytes of a wide (16 B per lane) coalesced
    stream; instead of assuming the factor 2, every pass CALIBRATES it: the child run first copies 1 GiB device to device (a
    transfer of known size through wide coalesced accesses, the kernel `copyBuffer` of the runtime), and bytes per count =
    2^30 / that kernel's counter.  The kernels of this path read their columns with 16 B per lane non-temporal loads (K2-K6;
    K5's chunk path too), i.e. the access pattern the calibration copy has.  A calibration outside [0.8, 2.5] for reads or
    [0.8, 1.3] for writes (or no copy kernel in the trace) falls back to the guide's 2 / 1 and says so.
    Each pass is a short child run of this script on the same workload (3 steps); launches of the full table are the ones whose
    counter is within 2x of the largest (the parity gate's launch is smaller).
    Returns (bytes, note, raw) or None (no rocprofv3, a failed pass: the caller falls back to profiles/traffic.json)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):  # no profiler, or this run is already a traced one
        return None
    vals, factor, how = {}, {}, {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="exon_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "p", "--output-format", "csv", "--", sys.executable,
                   os.path.abspath(__file__), "--workload", a.workload, "--rows", str(int(rows)), "--steps", "3", "--warmup", "1",
                   "--no-cpu-baseline", "--no-extras", "--no-pmc", "--calib-copy"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=180)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            rows_csv = list(csv.DictReader(open(files[0])))
            v = [float(row["Counter_Value"]) for row in rows_csv if PMC_KERNEL[a.workload] in row["Kernel_Name"] and row["Counter_Name"] == ctr]
            if not v:
                return None
            full = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v
            vals[ctr] = (sum(full) / len(full), len(full))
            cal = [float(row["Counter_Value"]) for row in rows_csv if "copyBuffer" in row["Kernel_Name"] and row["Counter_Name"] == ctr]
            cal = [x for x in cal if x > 0.25 * (1 << 20)]  # the 1 GiB copy (KiB counts of half a GiB and more), not the small ones
            default, lo, hi = (2.0, 0.8, 2.5) if ctr == "FETCH_SIZE" else (1.0, 0.8, 1.3)
            f = (1 << 20) / (sum(cal) / len(cal)) if cal else None
            if f is not None and lo <= f <= hi:
                factor[ctr], how[ctr] = f, f"calibrated on a 1 GiB device-to-device copy in the same pass ({f:.4f} bytes per counted byte)"
            else:
                factor[ctr], how[ctr] = default, (f"the guide's {default:g} (calibration copy " + ("not found in the trace" if f is None else f"gave {f:.3f}: out of range") + ")")
        except Exception:
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    b = int(factor["FETCH_SIZE"] * vals["FETCH_SIZE"][0] * 1024 + factor["WRITE_SIZE"] * vals["WRITE_SIZE"][0] * 1024)
    raw = {"FETCH_SIZE_KiB": round(vals["FETCH_SIZE"][0], 1), "WRITE_SIZE_KiB": round(vals["WRITE_SIZE"][0], 1),
           "fetch_bytes_per_count": round(factor["FETCH_SIZE"], 4), "write_bytes_per_count": round(factor["WRITE_SIZE"], 4)}
    note = (f"measured in this run: two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE, each with --kernel-trace only) of a 3-step child "
            f"run, {vals['FETCH_SIZE'][1]} full launches averaged; read bytes = FETCH_SIZE KiB x 1024 x {factor['FETCH_SIZE']:.4f} ({how['FETCH_SIZE']}), "
            f"write bytes = WRITE_SIZE KiB x 1024 x {factor['WRITE_SIZE']:.4f} ({how['WRITE_SIZE']}); raw counters in roofline.traffic_raw")
    return b, note, raw


def time_config(ctx, kind, rows, steps=50, warmup=5):
    """A config at the size BASELINE.json states it (c2 @ 1e7, c3 @ 1e8): (ms per step, kernel ms, roofline fraction)."""
    wl = Workload(ctx, kind, rows, 0, rows)
    for _ in range(warmup):
        wl.run()
    torch.cuda.synchronize()
    # ONE event pair around the K back-to-back steps: at these sizes a step is 20-250 us of GPU time and per-step event
    # records would be a visible share of it; kernel_ms = GPU time between the two events / K (gaps between steps included)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        wl.run()
    e1.record()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    kms = e0.elapsed_time(e1) / steps
    gbs = rows * BYTES_PER_ROW[kind] / (kms * 1e-3) / 1e9
    return {"rows": rows, "ms_per_step": round(ms, 4), "kernel_ms": round(kms, 4), "Mrows_per_s": round(rows / ms / 1e3, 1),
            "achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}


def h2d_inclusive(ctx, n=64_000_000):
    """PCIe-inclusive rate of the config-4 path (never `value`): host Arrow batches in pageable memory ->
    exon_hip_stream_push (staging copy into pinned memory on a few cores, async H2D, double-buffered) -> fused kernel, for
    4 Mi-row batches and the reference's 8192-row batches.  Driven by the native producer tools/bin/measure_h2d_native (a
    Python producer spends ~10 us per push in pyarrow export + ctypes and would measure itself)."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "bin", "measure_h2d_native")
    out = {}
    time.sleep(0.3)  # (the CPU baseline has just used the container's whole CFS quota: let the 100 ms period roll over)
    for name, batch in (("batch_4Mi_rows", 4 << 20), ("batch_8192_rows", 8192)):
        r = subprocess.run([exe, str(n), str(batch)], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-400:])
        out[name] = json.loads(r.stdout.strip().splitlines()[-1])
    out["note"] = ("native producer, one stream reused across repetitions (steady state of a partition); staging memcpy + H2D + "
                   "kernel; reported, never `value`")
    return out


def self_launch(a):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks ourselves with the driver's own command line
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...) and pass its output and exit code through.  The
    line printed then says n_gpus = N, or the run fails: `--gpus 8` can never come back as a 1-GPU number."""
    import socket
    import subprocess
    share = os.environ.get("EXON_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < 1:
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    if have < a.gpus and not share:
        raise SystemExit(f"--gpus {a.gpus} but this node exposes {have} GPU(s) (EXON_BENCH_SHARE_GPU=1 lets all ranks share cuda:0 "
                         f"for launcher tests; it measures nothing)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    a = parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # one process per GPU.  EXON_BENCH_SHARE_GPU=1 is a test hook for 1-GPU boxes: all ranks share cuda:0 and the
    # collective runs over gloo (RCCL refuses two ranks on one device); it exercises the launcher path, not xGMI.
    share = os.environ.get("EXON_BENCH_SHARE_GPU") == "1"
    device = 0 if share else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))

    import exon_amd
    from exon_amd.distributed import NativeComm, merge_state, shard_rows
    ctx = exon_amd.Context(device)
    if a.calib_copy:  # a transfer of KNOWN size for the PMC pass around this run: 1 GiB read, 1 GiB written, wide coalesced accesses
        cs_, cd_ = torch.empty(1 << 30, dtype=torch.uint8, device=f"cuda:{device}"), torch.empty(1 << 30, dtype=torch.uint8, device=f"cuda:{device}")
        cs_.fill_(1)
        torch.cuda.synchronize()
        cd_.copy_(cs_)
        torch.cuda.synchronize()
        del cs_, cd_
    # kernels, events and collectives all go on ONE explicit (non-default) HIP stream
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    if a.scaling == "strong":
        n_total = int(a.rows)
        lo, hi = shard_rows(n_total, rank, world)
    else:
        n_total = int(a.rows) * world
        lo, hi = rank * int(a.rows), (rank + 1) * int(a.rows)
    rows = hi - lo
    if a.groups != 5 and a.workload != "c4":
        raise SystemExit("--groups applies to --workload c4")
    if a.groups < 1:
        raise SystemExit("--groups must be >= 1")
    wl = Workload(ctx, a.workload, rows, lo, n_total, groups=a.groups, group_dist=a.group_dist)
    V = wl.state.numel()
    merged = torch.zeros_like(wl.state) if world > 1 else wl.state
    gathered = torch.zeros(world * V, dtype=torch.int64, device=wl.state.device) if world > 1 else None

    # ---- the merge across GPUs: pick the path outside the timed region --------------------------------------------
    merge_path = "none (1 GPU)"
    native = None
    hung_native_init = False
    if world > 1:
        wl.run()
        ref = merge_state(wl.state, wl.n_i64, gathered, merged, ctx=ctx).clone()  # also brings torch's communicator up
        merge_path = "torch.distributed all_gather_into_tensor + exon_hip_fold_states"
        if a.merge in ("auto", "native") and not share:
            ok = 1
            try:
                # ncclCommInitRank bootstraps over sockets: bound it, so that a rank that cannot join costs a minute and the
                # torch path, not the run (a thread stuck in native code cannot be cancelled; the process then leaves via os._exit)
                box = {}

                def _init():
                    try:
                        torch.cuda.set_device(device)
                        box["comm"] = NativeComm(ctx)
                    except Exception as e:  # noqa: BLE001
                        box["err"] = e
                th = threading.Thread(target=_init, daemon=True)
                th.start()
                th.join(float(os.environ.get("EXON_BENCH_NATIVE_INIT_TIMEOUT", 120)))
                if th.is_alive():
                    hung_native_init = True
                    raise TimeoutError("exon_hip_rccl_comm_init did not return in time")
                if "err" in box:
                    raise box["err"]
                native = box["comm"]
                merged.zero_()
                native.merge(wl.state, wl.n_i64, gathered, merged)
                torch.cuda.synchronize()
                ok = int(torch.equal(merged, ref))
            except Exception as e:  # noqa: BLE001 -- any failure means: use torch's collective
                ok = 0
                if a.merge == "native":
                    raise
                print(f"[rank {rank}] native RCCL merge unavailable ({e}); using torch.distributed", file=sys.stderr)
            flag = torch.tensor([ok], dtype=torch.int64, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # every rank takes the same path
            if int(flag.item()) == 1:
                merge_path = "exon_hip_merge_states: ncclAllGather on the kernels' stream + exon_hip_fold_states"
            else:
                native = None

    rccl_ranks = native.count()[0] if native is not None else (world if (world > 1 and not share) else None)

    def merge():
        if world == 1:
            return
        if native is not None:
            native.merge(wl.state, wl.n_i64, gathered, merged)
        else:
            merge_state(wl.state, wl.n_i64, gathered, merged, ctx=ctx)

    for _ in range(a.warmup):
        wl.run()
        merge()
    torch.cuda.synchronize()

    # What THIS box streams: a bare read (same grid, same 16 B/lane non-temporal loads, no predicate / aggregate) over the SAME
    # resident columns, before and after the timed steps (rank 0's figure is reported; every rank runs it so that the ranks stay
    # in step).  roofline.frac is against the spec peak; frac_of_box_ceiling is against what this box's HBM delivered just now.
    probe_ptrs, probe_bytes = wl.probe_buffers()
    probe = []

    def run_probe():
        if probe_ptrs:
            torch.cuda.synchronize()
            probe.append(ctx.read_probe(probe_ptrs, probe_bytes, reps=a.probe_reps, stream=torch.cuda.current_stream().cuda_stream))
    run_probe()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    ev_m = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)] if world > 1 else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        ev[i][0].record()
        wl.run()  # the hot path's kernel only (one launch; c5: offsets scan + main + fold)
        ev[i][1].record()
        merge()   # AggregateExec(Final) across GPUs: one all-gather over RCCL/xGMI + fixed-order fold
        if ev_m is not None:
            ev_m[i].record()
    torch.cuda.synchronize()
    t_done = time.perf_counter()  # CLOCK_MONOTONIC: one clock for all ranks of a node
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    run_probe()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    kern_ms = torch.tensor([sum(s.elapsed_time(e) for s, e in ev) / a.steps], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(kern_ms, op=dist.ReduceOp.MAX)
    elapsed, kern_ms = float(elapsed.item()), float(kern_ms.item())
    # N > 1: what every rank saw, so that a scaling figure can be split into shard kernel / collective / skew (DESIGN section 7's
    # prediction): mean kernel time, mean GPU time from the end of the kernel to the end of the merge (all-gather + fold: it
    # INCLUDES the wait for the slowest rank's kernel, which is what the collective costs this rank), and when each rank started
    # and finished its K steps on the node's monotonic clock
    per_rank = None
    if world > 1:
        mine = torch.tensor([sum(s.elapsed_time(e) for s, e in ev) / a.steps,
                             sum(ev[i][1].elapsed_time(ev_m[i]) for i in range(a.steps)) / a.steps * 1e3,
                             t0, t_done], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rows_ = [x.cpu().tolist() for x in allr]
        start0 = min(r[2] for r in rows_)
        per_rank = {"kernel_ms": [round(r[0], 4) for r in rows_], "merge_us": [round(r[1], 1) for r in rows_],
                    "start_skew_us": [round((r[2] - start0) * 1e6, 1) for r in rows_],
                    "finish_skew_us": [round((r[3] - min(q[3] for q in rows_)) * 1e6, 1) for r in rows_],
                    "note": "merge_us = GPU time from the end of this rank's kernel to the end of its merge (all-gather + fold), i.e. it "
                            "includes waiting for the slowest rank's kernel; skews on the node's CLOCK_MONOTONIC: start = leaving the "
                            "barrier in front of the timed steps, finish = this rank's last step done"}

    final = merged.cpu()
    counts = final[:wl.n_i64].numpy()
    sums = final[wl.n_i64:].view(torch.float64).numpy() if wl.n_f64 else None

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = n_total / (elapsed / a.steps) / 1e6
        bpr = BYTES_PER_ROW[a.workload]
        achieved = rows * bpr / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "Mrows/sec filter+agg on 1B-row synthetic VCF; achieved HBM GB/s vs peak"
            if a.workload == "c4" else f"Mrows/sec filter+agg ({a.workload})",
            "value": round(value, 1), "unit": "Mrows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": DTYPE[a.workload], "data": "synthetic",
            "config": {"workload": WORKLOAD[a.workload] + (f" -- with {a.groups} synthetic GROUP BY keys ({a.group_dist})" if a.groups != 5 else ""),
                       "rows_total": n_total, "rows_per_gpu": rows,
                       "sharding": f"rank k owns the contiguous row range (file split) [k N/{world}, (k+1) N/{world})",
                       "reduce": merge_path, "state_bytes": V * 8,
                       "rccl_ranks": rccl_ranks,  # ncclCommCount of the communicator the merge ran on (None: 1 GPU / gloo test hook)
                       "rccl_ranks_source": ("ncclCommCount" if native is not None else
                                             "torch.distributed world size (RCCL backend)" if rccl_ranks else None),
                       "bytes_per_row": bpr, "arithmetic": ARITH[a.workload],
                       "generator": GENERATOR_NOTE.get(a.workload, "counter-based generator of DESIGN.md section 5")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel_ms": round(kern_ms, 4),
                         "note": "achieved = rows_per_gpu x bytes_per_row / mean HIP-event time of one launch (max over ranks).  A launch is "
                                 "ONE kernel: the last workgroup to finish folds the per-workgroup records and writes the state "
                                 "(no finalize launch, no zeroing pass); c5: offsets scan + main + fold"},
        }
        if per_rank is not None:
            out["per_rank"] = per_rank
        if probe:
            gb = [p[0] for p in probe]
            ceil = max(gb)
            out["roofline"]["box_read_ceiling_GBps"] = round(ceil, 1)
            out["roofline"]["frac_of_box_ceiling"] = round(achieved / ceil, 4)
            out["roofline"]["box_read_probe"] = {
                "before_GBps": round(gb[0], 1), "after_GBps": round(gb[-1], 1), "reps": a.probe_reps,
                "frac_of_peak": round(ceil / HBM_PEAK_GBS, 4),
                "note": "exon_hip_read_probe: bare streaming read (the fused kernels' persistent grid and 16 B/lane non-temporal loads, "
                        "four integer adds per load, no predicate / aggregate) over the same resident value columns walked in "
                        "lock-step, before and after the timed steps; ceiling = the better of the two; the kernel additionally "
                        "reads the validity bitmaps (0.25 B/row in c4), which the probe leaves out"}
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        live = measure_traffic(a, rows) if (world == 1 and not a.no_pmc and a.groups == 5) else None
        if live is not None:
            out["roofline"]["traffic"], out["roofline"]["traffic_source"], out["roofline"]["traffic_raw"] = live
        elif os.path.exists(traffic_file):
            try:
                t = json.load(open(traffic_file)).get(a.workload)
                if t and int(t.get("rows", 0)) == rows and a.groups == 5:
                    out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = (f"profiles/traffic.json (builder's rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                         f"passes of round {t.get('round', '?')}, NOT measured in this run)")
            except Exception:
                pass
        if a.groups != 5:
            # no C-oracle generator for synthetic keys: the checker is a plain torch fp64 statement of the query on the
            # first rows of the same resident table
            m = int(min(rows, 64_000_000))
            chk = Workload(ctx, "c4", m, lo, n_total, groups=a.groups, group_dist=a.group_dist)
            chk.run()
            torch.cuda.synchronize()
            cnn, crow, sm = torch_reference_c4(chk)
            G = a.groups
            if not (torch.equal(chk.counts[:G], cnn) and torch.equal(chk.counts[G:], crow)):
                raise SystemExit("PARITY FAILURE: counts vs torch reference")
            if not torch.allclose(chk.sums, sm, rtol=1e-9, atol=0):
                raise SystemExit("PARITY FAILURE: sums vs torch reference")
            out["parity"] = f"bit-exact counts, sums within 1e-9 rel. vs a torch fp64 reference on rows [{lo},{lo + m})"
            # tiers of the > 8-key kernel (kernels.hip: K4_OVF_REGS = 4 register ids, K4_LDS_GROUPS = 4100 ids up to the LDS
            # table's end, the rest partitioned by id range and aggregated range by range); <= 8 keys: all in registers
            reg = G if G <= 8 else 4
            out["config"]["groups"] = {"n": G, "dist": a.group_dist, f"rows_in_registers_ids_lt_{reg}": int(crow[:reg].sum()),
                                       "rows_in_lds_table_ids_4_to_4099": int(crow[reg:4100].sum()) if G > 8 else 0,
                                       "rows_partitioned_ids_ge_4100": int(crow[4100:].sum()) if G > 8 else 0,
                                       "of_rows_passing": int(crow.sum()), "sample_rows": m}
            del chk
        elif not a.no_cpu_baseline and world == 1:
            sample = int(min(a.cpu_sample_rows, rows))
            if a.workload == "c5":
                sample = min(sample, 8_000_000)
            if a.workload == "c6":
                sample = min(sample, 32_000_000)
            base, (oc, os_) = cpu_baseline(a.workload, sample, n_total, a.cpu_reps)
            out["cpu_baseline"] = base
            # parity gate: the GPU path over the same sample rows must reproduce the oracle
            chk = Workload(ctx, a.workload, sample, 0, n_total)
            chk.run()
            torch.cuda.synchronize()
            gc = chk.counts.cpu().numpy()
            if not np.array_equal(gc, oc):
                raise SystemExit(f"PARITY FAILURE: counts {gc} vs oracle {oc}")
            if os_ is not None and not np.allclose(chk.sums.cpu().numpy(), os_, rtol=1e-6, atol=0):
                raise SystemExit("PARITY FAILURE: sums")
            out["parity"] = f"bit-exact counts, sums within 1e-6 rel. vs oracle on rows [0,{sample})"
            del chk
        if world == 1 and not a.no_extras and a.workload == "c4" and a.groups == 5:
            extras = {}
            try:
                extras["configs_at_stated_size"] = {"c2_1e7_rows": time_config(ctx, "c2", 10_000_000),
                                                    "c3_1e8_rows": time_config(ctx, "c3", 100_000_000),
                                                    "c4_shard_of_8_125e6_rows": time_config(ctx, "c4", 125_000_000)}
                extras["h2d_inclusive"] = h2d_inclusive(ctx)
            except Exception as e:  # noqa: BLE001 -- side measurements never cost the headline line
                extras["error"] = repr(e)
            out["extras"] = extras
        if a.workload == "c4" and a.groups > 64:
            out["result"] = {"groups_observed": int((counts[a.groups:] > 0).sum()), "rows_passing": int(counts[a.groups:].sum())}
        elif a.workload == "c4":
            G = a.groups
            out["result"] = {"filter_rows": counts[G:].tolist(),
                             "avg_qual": [float(sums[g] / counts[g]) if counts[g] else None for g in range(G)]}
        else:
            out["result"] = {"counts": counts.tolist()}
        print(json.dumps(out))
    ctx.sync()
    if native is not None:
        native.close()
    if world > 1:
        dist.destroy_process_group()
    if hung_native_init:  # a thread is still inside ncclCommInitRank: leave without waiting for it
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
