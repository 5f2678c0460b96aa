#!/usr/bin/env python3
"""bench.py -- headline benchmark: filter + GROUP BY aggregate over a synthetic VCF (BASELINE.json config 4).

    SET exon.vcf_parse_info = true;
    SELECT filter, AVG(qual), COUNT(*) FROM v WHERE info."AF" > 0.01 GROUP BY filter

One "step" = one full pass of the fused filter+aggregate kernel over this rank's HBM-resident shard
(default 1e9 rows per GPU: weak scaling, one file split per GPU), state zeroing included, followed -- when
N > 1 -- by the RCCL all-reduce of the partial aggregate state (5 x {f64 sum, 2 x i64 count} = 120 B).
Launch: `python bench.py --gpus 1` or
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N --steps K --warmup W`.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (imported before the HIP library so both share one HIP runtime)
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
BYTES_PER_ROW = {"c4": 12.25, "c2": 12.0, "c3": 9.25, "c5": 104.0, "c6": 20.375}  # algorithmic bytes/row, SURVEY.md section 8(d)
SEED = {"c2": 2, "c3": 3, "c4": 4, "c5": 5, "c6": 6}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=float, default=1e9, help="rows per GPU (weak scaling)")
    ap.add_argument("--workload", default="c4", choices=["c4", "c2", "c3", "c5", "c6"])
    ap.add_argument("--cpu-sample-rows", type=float, default=128e6)
    ap.add_argument("--cpu-reps", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


class Workload:
    """Device-resident synthetic shard + one launch of the hot path on torch's current stream."""

    def __init__(self, ctx, kind, rows, row0, n_total):
        self.ctx, self.kind, self.n = ctx, kind, rows
        dev = torch.device("cuda", torch.cuda.current_device())
        s = torch.cuda.current_stream().cuda_stream
        lib, h = ctx.lib, ctx.h
        nb = (rows + 7) // 8 + 64
        if kind == "c4":
            self.af = torch.empty(rows, dtype=torch.float32, device=dev)
            self.qual = torch.empty(rows, dtype=torch.float32, device=dev)
            self.fid = torch.empty(rows, dtype=torch.int32, device=dev)
            self.av = torch.zeros(nb, dtype=torch.uint8, device=dev)
            self.qv = torch.zeros(nb, dtype=torch.uint8, device=dev)
            ctx._check(lib.exon_hip_gen_c4(h, s, SEED["c4"], row0, row0 + rows, self.af.data_ptr(), self.av.data_ptr(),
                                           self.qual.data_ptr(), self.qv.data_ptr(), self.fid.data_ptr()))
            self.G = 5
            self.counts = torch.zeros(2 * self.G, dtype=torch.int64, device=dev)
            self.sums = torch.zeros(self.G, dtype=torch.float64, device=dev)
        elif kind == "c2":
            self.chrom = torch.empty(rows, dtype=torch.int32, device=dev)
            self.pos = torch.empty(rows, dtype=torch.int64, device=dev)
            ctx._check(lib.exon_hip_gen_c2(h, s, SEED["c2"], n_total, row0, row0 + rows, self.chrom.data_ptr(),
                                           self.pos.data_ptr()))
            self.counts = torch.zeros(1, dtype=torch.int64, device=dev)
            self.sums = None
        elif kind == "c6":
            self.ref = torch.empty(rows, dtype=torch.int32, device=dev)
            self.start = torch.empty(rows, dtype=torch.int64, device=dev)
            self.end = torch.empty(rows, dtype=torch.int64, device=dev)
            self.rv = torch.zeros(nb, dtype=torch.uint8, device=dev)
            self.pv = torch.zeros(nb, dtype=torch.uint8, device=dev)
            ctx._check(lib.exon_hip_gen_c6(h, s, SEED["c6"], row0, row0 + rows, self.ref.data_ptr(), self.rv.data_ptr(),
                                           self.start.data_ptr(), self.end.data_ptr(), self.pv.data_ptr()))
            self.counts = torch.zeros(1, dtype=torch.int64, device=dev)
            self.sums = None
        elif kind == "c3":
            self.flag = torch.empty(rows, dtype=torch.int32, device=dev)
            self.mapq = torch.empty(rows + 64, dtype=torch.uint8, device=dev)
            self.ref = torch.empty(rows, dtype=torch.int32, device=dev)
            self.mv = torch.zeros(nb, dtype=torch.uint8, device=dev)
            self.rv = torch.zeros(nb, dtype=torch.uint8, device=dev)
            ctx._check(lib.exon_hip_gen_c3(h, s, SEED["c3"], row0, row0 + rows, self.flag.data_ptr(), self.mapq.data_ptr(),
                                           self.mv.data_ptr(), self.ref.data_ptr(), self.rv.data_ptr()))
            self.R = 25
            self.counts = torch.zeros(self.R + 1, dtype=torch.int64, device=dev)
            self.sums = None
        elif kind == "c5":
            # FASTQ quality strings, L = 100.  Arrow Utf8 has int32 offsets, so the shard is a sequence of
            # batches of <= 16 Mi reads (1.6 GB of quality bytes each); the histogram state accumulates.
            self.L = 100
            self.batch = min(rows, 16 << 20)
            self.bytes = torch.empty(rows * self.L + 64, dtype=torch.uint8, device=dev)
            self.off = torch.empty(self.batch + 1, dtype=torch.int32, device=dev)
            for b0 in range(0, rows, self.batch):
                nb_ = min(self.batch, rows - b0)
                ctx._check(lib.exon_hip_gen_c5(h, s, SEED["c5"], row0 + b0, row0 + b0 + nb_, self.L, self.off.data_ptr(),
                                               self.bytes.data_ptr() + b0 * self.L))
            if rows % self.batch:  # offsets of a full batch (the last generator call wrote a shorter one)
                ctx._check(lib.exon_hip_gen_c5(h, s, SEED["c5"], 0, self.batch, self.L, self.off.data_ptr(),
                                               self.bytes.data_ptr()))
                ctx._check(lib.exon_hip_gen_c5(h, s, SEED["c5"], row0, row0 + self.batch, self.L, self.off.data_ptr(),
                                               self.bytes.data_ptr()))
            self.counts = torch.zeros(self.L * 256, dtype=torch.int64, device=dev)
            self.sums = None
        torch.cuda.synchronize()

    def launch(self):
        self.zero()
        self.run()

    def zero(self):
        """Aggregation state of one step (part of the step, not of the kernel: outside the HIP-event bracket)."""
        self.counts.zero_()
        if self.sums is not None:
            self.sums.zero_()

    def run(self):
        from exon_amd.engine import _col
        import ctypes as C
        ctx, n = self.ctx, self.n
        s = torch.cuda.current_stream().cuda_stream
        if self.kind == "c4":
            c0 = _col(self.af.data_ptr(), self.av.data_ptr(), None, n)
            c1 = _col(self.qual.data_ptr(), self.qv.data_ptr(), None, n)
            c2 = _col(self.fid.data_ptr(), None, None, n)
            ctx._check(ctx.lib.exon_hip_cmp_avg_by_group(ctx.h, s, C.byref(c0), C.byref(c1), C.byref(c2), n, 0.01, 0,
                                                         self.G, self.counts.data_ptr(), self.sums.data_ptr()))
        elif self.kind == "c2":
            c0, c1 = _col(self.chrom.data_ptr(), None, None, n), _col(self.pos.data_ptr(), None, None, n)
            ctx._check(ctx.lib.exon_hip_region_count(ctx.h, s, C.byref(c0), C.byref(c1), n, 6, 50000000, 100000000,
                                                     self.counts.data_ptr()))
        elif self.kind == "c6":
            c0 = _col(self.ref.data_ptr(), self.rv.data_ptr(), None, n)
            c1 = _col(self.start.data_ptr(), self.pv.data_ptr(), None, n)
            c2 = _col(self.end.data_ptr(), self.pv.data_ptr(), None, n)
            ctx._check(ctx.lib.exon_hip_overlap_count(ctx.h, s, C.byref(c0), C.byref(c1), C.byref(c2), n, 6, 50000000, 100000000,
                                                      self.counts.data_ptr()))
        elif self.kind == "c5":
            for b0 in range(0, n, self.batch):
                nb_ = min(self.batch, n - b0)
                c0 = _col(self.bytes.data_ptr() + b0 * self.L, None, self.off.data_ptr(), nb_)
                ctx._check(ctx.lib.exon_hip_qual_pos_hist(ctx.h, s, C.byref(c0), nb_, self.L, self.counts.data_ptr()))
        else:
            c0 = _col(self.flag.data_ptr(), None, None, n)
            c1 = _col(self.mapq.data_ptr(), self.mv.data_ptr(), None, n)
            c2 = _col(self.ref.data_ptr(), self.rv.data_ptr(), None, n)
            ctx._check(ctx.lib.exon_hip_flag_mapq_group_count(ctx.h, s, C.byref(c0), C.byref(c1), C.byref(c2), n, 1284,
                                                              0, 30, self.R, self.counts.data_ptr()))


def cpu_baseline(kind, sample_rows, n_total, reps):
    """The CPU restatement of the Exon/DataFusion plan (oracle/exon_oracle.c) on a bounded sample of the
    SAME synthetic rows [0, sample_rows), all host cores; returns (result dict, oracle outputs)."""
    from oracle import Oracle
    orc = Oracle()
    n = int(sample_rows)
    secs, mat = [], []
    if kind == "c4":
        af, av, q, qv, fid = orc.gen_c4(SEED["c4"], 0, n)
        for _ in range(reps):
            s, cn, cr, t = orc.c4_cmp_avg_by_group(af, av, q, qv, fid, orc.c4_filters(), 0.01, ">")
            secs.append(t.seconds_exec), mat.append(t.seconds_materialize)
        out = (np.concatenate([cn, cr]), s)
    elif kind == "c2":
        c, p = orc.gen_c2(SEED["c2"], n_total, 0, n)
        for _ in range(reps):
            r, t = orc.c2_region_count(c, p, orc.c2_contigs(), "7:50000000-100000000")
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
        off, data = orc.gen_c5(SEED["c5"], 0, n, 100)
        for _ in range(reps):
            hcpu, t = orc.c5_qual_pos_hist(off, data, 100)
            secs.append(t.seconds_exec), mat.append(t.seconds_materialize)
        out = (hcpu.reshape(-1), None)
    else:
        f, mq, mv, ref, rv = orc.gen_c3(SEED["c3"], 0, n)
        for _ in range(reps):
            cnt, t = orc.c3_flag_mapq_group_count(f, mq, mv, ref, rv, orc.c3_refs(), 1284, 0, 30)
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
           "single_thread_value": one,
           "sample": f"rows [0,{n}) of the same synthetic table as 8192-row Arrow-layout batches (Utf8/List<Utf8> keys), "
                     f"{t.threads} partitions = host cores; median of {reps} runs, {best:.3f}s exec each "
                     f"(+{float(np.median(mat)):.2f}s untimed Arrow-layout build); total CPU work "
                     f"~{sum(secs) * t.threads:.0f} core-seconds"}
    return res, out


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
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
    ctx = exon_amd.Context(device)
    # kernels, state zeroing, events and collectives all go on ONE explicit (non-default) HIP stream
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    rows = int(a.rows)
    n_total = rows * world
    wl = Workload(ctx, a.workload, rows, rank * rows, n_total)

    from exon_amd.distributed import all_reduce_state

    def step():
        wl.launch()
        all_reduce_state(wl.counts, wl.sums)  # AggregateExec(Final) across GPUs: RCCL all-reduce over xGMI

    if world > 1:  # bring the communicator up outside the timed region even with --warmup 0
        dist.all_reduce(torch.zeros(1, dtype=torch.int64, device="cuda"))
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        wl.zero()
        ev[i][0].record()
        wl.run()  # the hot path's kernels only: main + finalize (c5: per batch, offsets scan + main + finalize)
        ev[i][1].record()
        all_reduce_state(wl.counts, wl.sums)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    kern_ms = torch.tensor([sum(s.elapsed_time(e) for s, e in ev) / a.steps], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(kern_ms, op=dist.ReduceOp.MAX)
    elapsed, kern_ms = float(elapsed.item()), float(kern_ms.item())

    counts = wl.counts.cpu().numpy()
    sums = wl.sums.cpu().numpy() if wl.sums is not None else None

    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = n_total / (elapsed / a.steps) / 1e6
        bpr = BYTES_PER_ROW[a.workload]
        achieved = rows * bpr / (kern_ms * 1e-3) / 1e9
        out = {
            "metric": "Mrows/sec filter+agg on 1B-row synthetic VCF; achieved HBM GB/s vs peak"
            if a.workload == "c4" else f"Mrows/sec filter+agg ({a.workload})",
            "value": round(value, 1), "unit": "Mrows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"c4": "f64", "c2": "int64", "c3": "int64", "c5": "u8", "c6": "int64"}[a.workload], "data": "synthetic",
            "config": {"workload": {"c4": "config 4: 1B-row synthetic VCF, WHERE info.AF > 0.01, AVG(qual), COUNT(*) GROUP BY filter",
                                    "c2": "config 2: synthetic VCF, chrom='7' AND pos in [5e7,1e8], COUNT(*)",
                                    "c3": "config 3: synthetic BAM, flag&1284=0 AND mapq>=30, COUNT(*) GROUP BY reference",
                                    "c5": "config 5: synthetic FASTQ (L=100), per-position 256-bin quality histogram; rows = reads",
                                    "c6": "synthetic alignments, bam_region_filter('<ref 7>:50000000-100000000', reference, start, end), COUNT(*)"}[a.workload],
                       "rows_per_gpu": rows, "rows_total": n_total, "sharding": "one contiguous row range (file split) per GPU",
                       "reduce": "RCCL all-reduce of partial state" if world > 1 else "none (1 GPU)",
                       "bytes_per_row": bpr,
                       "arithmetic": {"c4": "f32 columns compared as totalOrder i32 keys, f64 sums, i64 counts",
                                      "c2": "i32 / i64 compares, i64 count", "c3": "i32 mask compare, u8 compare, i64 counts",
                                      "c5": "u8 bytes, u32 LDS counters folded into i64", "c6": "i32 / i64 compares, i64 count"}[a.workload]},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel_ms": round(kern_ms, 4),
                         "note": "achieved = rows_per_gpu x bytes_per_row / mean HIP-event time of one launch (main + finalize kernels; the state zeroing of the step is outside the bracket)"},
        }
        traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(traffic_file):
            try:
                t = json.load(open(traffic_file)).get(a.workload)
                if t and int(t.get("rows", 0)) == rows:
                    out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
            except Exception:
                pass
        if not a.no_cpu_baseline and world == 1:
            sample = int(min(a.cpu_sample_rows, rows))
            if a.workload == "c5":
                sample = min(sample, 8_000_000)
            if a.workload == "c6":
                sample = min(sample, 32_000_000)
            base, (oc, os_) = cpu_baseline(a.workload, sample, n_total, a.cpu_reps)
            out["cpu_baseline"] = base
            # parity gate: the GPU path over the same sample rows must reproduce the oracle
            chk = Workload(ctx, a.workload, sample, 0, n_total)
            chk.launch()
            torch.cuda.synchronize()
            gc = chk.counts.cpu().numpy()
            if not np.array_equal(gc, oc):
                raise SystemExit(f"PARITY FAILURE: counts {gc} vs oracle {oc}")
            if os_ is not None and not np.allclose(chk.sums.cpu().numpy(), os_, rtol=1e-6, atol=0):
                raise SystemExit("PARITY FAILURE: sums")
            out["parity"] = f"bit-exact counts, sums within 1e-6 rel. vs oracle on rows [0,{sample})"
        if a.workload == "c4":
            G = 5
            out["result"] = {"filter_rows": counts[G:].tolist(),
                             "avg_qual": [float(sums[g] / counts[g]) if counts[g] else None for g in range(G)]}
        else:
            out["result"] = {"counts": counts.tolist()}
        print(json.dumps(out))
    ctx.sync()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
