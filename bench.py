#!/usr/bin/env python
"""Headline benchmark: short-read score_chain polishing throughput (BASELINE.json metric) on the metric's own configuration.

Workload (default c5_3gb_30x = the configuration BASELINE.json's metric is quoted on: a 3 Gb human-scale draft at 30x; it fits one
MI355X): a FIXED synthetic draft (contig lengths log-uniform) + 30x simulated 2x150 bp PE reads = 600 M records.
The contigs are dealt longest-first over the N ranks (contigs are independent units: strong scaling of one fixed draft, no
data-path collective; reference: source/lib/nextpolish1.py:181-189,223-224) and packed into batches of <= 260 Mb.

One "step" = one full score_chain pass over the WHOLE draft with every record batch RESIDENT IN HBM when the timed region starts
(SURVEY.md 8d's timing scope 1; since round 6 -- rounds 1-5 timed the PCIe-inclusive pass and reported the resident one beside it):
every kernel of np1_device.hip's launch sequence over all batches on the device lanes (`value`).  Beside it, in the same JSON line:
  streamed          the PCIe-inclusive rate of the same pass: decoded records in pinned host memory -> async H2D -> kernels -> D2H of the
                    polished strings, double-buffered on the lanes (bound by the H2D of the record stream; never `value`)
  e2e_from_files    FASTA + sorted BAM on disk (page cache) -> polished FASTA at the full size, on a BAM with Illumina-like binned
                    base qualities: cold process of the CLI and a warm process (scope 2); the BAM's bytes per record; on a one-batch
                    slice the same with no qualities (9:1) and with incompressible ones
  parity            contigs of this very run compared with the CPU oracle (md5 of the polished strings)
  roofline          dominant kernel: algorithmic bytes per launch / HIP-event time per launch vs the 8 TB/s HBM peak, and
                    the PMC traffic per launch (two rocprofv3 --pmc passes)
  cpu_baseline      the compiled reference (oracle/_ref/nextpolish1) on this box's host cores, one process per core like -p N,
                    and one core alone, on a bounded sample of the same shape (rank 0, N=1 only)
  lgs               the long-read path (lib/nextpolish2.so ctg_cns_core) with its own roofline and cpu_baseline
  snp_phase         task 3 (np1_batch_snp_phase): diploid draft, short + long reads resident in HBM, with the compiled reference beside it (N=1 only)
"""
import argparse
import json
import math
import os
import random
import shutil
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
WORKLOADS = {
    # name: (total draft bp, depth, min contig, max contig, batch bp)      -- SURVEY.md 8d synthetic shapes
    "c3_100mb_30x": (100000000, 30.0, 50e3, 5e6, 13000000),
    "c2_5mb_50x": (5000000, 50.0, 1e6, 2.5e6, 13000000),
    "c5_3gb_30x": (3000000000, 30.0, 2e6, 250e6, 260000000),
    "small_8mb_30x": (8000000, 30.0, 50e3, 2e6, 3000000),
}


def contig_lengths(total, lo, hi, seed=20250117 + 3):
    rng = random.Random(seed)
    lens, acc = [], 0
    while acc < total:
        L = int(math.exp(rng.uniform(math.log(lo), math.log(hi))))
        L = min(L, total - acc) if total - acc > lo else total - acc
        lens.append(L)
        acc += L
    return lens


def host_cores():
    n = len(os.sched_getaffinity(0))
    try:   # cgroup v2 CPU quota, when the box has one
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(depth, sample_mb, procs):
    """The compiled reference CLI (BAM + BGZF inflate -> FASTA): `procs` processes at once, one sample each (the reference's
    -p N model), then one process alone.  Falls back to the oracle port (1 core) when oracle/_ref did not travel."""
    from nextpolish_amd import _native as nat
    ref = os.path.join(ROOT, "oracle", "_ref", "nextpolish1")
    L = int(sample_mb * 1e6)
    if not os.path.exists(ref):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding as ob
        st = nat.Stream.synth([L], depth=depth, seed=424242)
        t0 = time.time()
        ob.score_chain(st, 0)
        dt = time.time() - t0
        return {"value": round(L / 1e6 / dt, 4), "unit": "Mbp/s", "cores": 1, "kind": "port",
                "sample": "%.1f Mb synthetic draft, %.0fx PE150, score_chain of the oracle port in memory, %.1f s" % (L / 1e6, depth, dt)}
    td = tempfile.mkdtemp(prefix="np1cpu_")
    try:
        def make(k):
            st = nat.Stream.synth([L], depth=depth, seed=424242 + k, prefix="s%dctg" % k)
            fa, bam = os.path.join(td, "s%d.fa" % k), os.path.join(td, "s%d.bam" % k)
            st.write_files(fa, bam, 1)
            st.close()
            return fa, bam
        with ThreadPoolExecutor(min(8, procs)) as ex:
            files = list(ex.map(make, range(procs)))
        t0 = time.time()
        ps = [subprocess.Popen([ref, "scorechain", fa, bam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for fa, bam in files]
        for p in ps:
            p.wait()
        dt_all = time.time() - t0
        t0 = time.time()
        subprocess.run([ref, "scorechain", files[0][0], files[0][1]], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        dt_one = time.time() - t0
    finally:
        shutil.rmtree(td, ignore_errors=True)
    return {"value": round(procs * L / 1e6 / dt_all, 4), "unit": "Mbp/s", "cores": procs, "kind": "reference",
            "one_core": round(L / 1e6 / dt_one, 4),
            "sample": "%d processes x %.1f Mb synthetic draft, %.0fx PE150, nextpolish1 scorechain, BAM(+BGZF inflate) -> FASTA, %.1f s; "
                      "one process alone %.1f s" % (procs, L / 1e6, depth, dt_all, dt_one)}


def rocprof_counter(cmd, ctr, env=None):
    """One rocprofv3 --pmc pass over `cmd`; returns {kernel_name: (avg value per dispatch, dispatches)}."""
    import sqlite3
    exe = shutil.which("rocprofv3")
    if not exe:
        raise RuntimeError("rocprofv3 not found")
    td = tempfile.mkdtemp(prefix="np1pmc_", dir="/tmp")
    try:
        subprocess.run([exe, "--pmc", ctr, "-d", td, "-o", "p", "--"] + cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=600, check=True, cwd="/tmp", env=dict(env or os.environ, TMPDIR="/tmp"))
        db = None
        for root, _d, files in os.walk(td):
            for f in files:
                if f.endswith(".db"):
                    db = os.path.join(root, f)
        c = sqlite3.connect(db)
        rows = c.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name=? group by kernel_name",
                         (ctr,)).fetchall()
        return {r[0]: (float(r[1]), int(r[2])) for r in rows}
    finally:
        shutil.rmtree(td, ignore_errors=True)


def rocprof_kernel_stats(cmd, env=None):
    """One rocprofv3 --kernel-trace --stats pass over `cmd`; returns [(kernel name, calls, total ns, average ns)] sorted by total time."""
    import csv
    exe = shutil.which("rocprofv3")
    if not exe:
        raise RuntimeError("rocprofv3 not found")
    td = tempfile.mkdtemp(prefix="np1kst_", dir="/tmp")
    try:
        subprocess.run([exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", td, "-o", "k", "--"] + cmd, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, timeout=600, check=True, cwd="/tmp", env=dict(env or os.environ, TMPDIR="/tmp"))
        rows = []
        for root, _d, files in os.walk(td):
            for f in files:
                if f.endswith("kernel_stats.csv"):
                    for r in csv.DictReader(open(os.path.join(root, f))):
                        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"])))
        rows.sort(key=lambda r: -r[2])
        return rows
    finally:
        shutil.rmtree(td, ignore_errors=True)


def pmc_traffic(args, kernel_substr):
    """HBM bytes of the dominant kernel per launch: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE do not fit one pass on
    gfx950) over a short child run of this script on the same workload.  Units/corrections per MI355X_MICROARCH.md (HBM
    section): both counters are KiB per dispatch; on gfx950 FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced
    reads, which is how this kernel stages its inputs, so the read side is doubled; WRITE_SIZE is taken as is."""
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            rows = rocprof_counter([sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", args.workload,
                                    "--steps", "1", "--warmup", "0", "--pmc-batches", str(args.pmc_batches)], ctr)
            hit = [v for k, v in rows.items() if kernel_substr in k]
            vals[ctr] = sum(a * n for a, n in hit) / max(1, sum(n for _a, n in hit))
        except Exception as e:   # profiling is best effort: the bench line stays valid without it
            return None, "pmc pass failed: %r" % (e,)
    fetch_b, write_b = vals["FETCH_SIZE"] * 1024.0, vals["WRITE_SIZE"] * 1024.0
    return {"bytes": int(2 * fetch_b + write_b), "fetch_size_kib_raw": round(vals["FETCH_SIZE"], 1),
            "write_size_kib_raw": round(vals["WRITE_SIZE"], 1), "correction": "2*FETCH_SIZE + WRITE_SIZE (gfx950)",
            "profiled": "child run of this script over the first %d batches of the workload (same seeds), every launch of the kernel averaged" % args.pmc_batches}, None


LGS_WORKER = r"""
import ctypes as C, os, resource, sys, time
sys.path.insert(0, %(root)r)
from nextpolish_amd import nextpolish2 as h
P = h.load_library(%(lib)r)
refs = P.read_ref(%(fa)r.encode(), None, 0)
cfg = P.ctg_cns_init(5000000, 1, 0, 0.8, 0.8, 0.8)
def once():
    n = 0
    for i in range(refs.contents.i):
        d = P.ctg_cns_core(cfg, C.byref(refs.contents.ref[i]), %(fofn)r.encode())
        n += sum(int(d.contents.data[k].len) for k in range(d.contents.i_m))
        P.free_consensus_trimed_data(d)
    return n
once()                                    # warm-up: HIP context + buffers in HBM
open(%(ready)r, "w").close()
while not os.path.exists(%(go)r): time.sleep(0.002)
t0 = time.time(); c0 = resource.getrusage(resource.RUSAGE_SELF)
bp = sum(once() for _ in range(%(calls)d))
c1 = resource.getrusage(resource.RUSAGE_SELF)
print(t0, time.time(), bp, c1.ru_utime + c1.ru_stime - c0.ru_utime - c0.ru_stime)
"""


def _short_kernel(name):
    """rocprofv3's demangled name without its argument list: 'void np2::(anonymous namespace)::k_x<4>(int*)' -> 'np2::k_x<4>'"""
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0].replace("void ", "").strip()


def lgs_roofline(worker_code, env, alg_bytes, with_pmc):
    """Per-window device time of the long-read path from the library's own stage clock (NP2_TIMING: stream-synchronised wall
    time per stage, second call = warm), the dominant stage, and the PMC traffic of ALL its kernels per window."""
    code = worker_code.replace("while not os.path.exists", "while False and os.path.exists")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, NP2_TIMING="1"))
    if p.returncode != 0:
        return {"error": p.stderr[-300:]}
    stages = {}
    n_windows = 0
    for line in p.stderr.splitlines():
        if line.startswith("[np2 window]") and "| ms:" in line:
            n_windows += 1
            toks = line.split("| ms:")[1].split()
            for k in range(0, len(toks) - 1, 2):
                stages[toks[k]] = stages.get(toks[k], 0.0) + float(toks[k + 1])
    if not stages:
        return {"error": "no stage lines"}
    last = {}
    for line in p.stderr.splitlines():          # the last window line = the warm call
        if line.startswith("[np2 window]") and "| ms:" in line:
            toks = line.split("| ms:")[1].split()
            last = {toks[k]: float(toks[k + 1]) for k in range(0, len(toks) - 1, 2)}
    dev = {k: v for k, v in last.items() if k != "download"}
    dom = max(dev, key=lambda k: dev[k])
    total_ms = sum(dev.values())
    out = {"bound": "hbm", "kernel": "stage '%s' of the window executor (np2_exec_hip.hip)" % dom,
           "achieved": round(alg_bytes / (dev[dom] * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(alg_bytes / (dev[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6), "kernel_ms": round(dev[dom], 3),
           "algorithmic_bytes_per_window": int(alg_bytes), "device_ms_per_window": round(total_ms, 2),
           "achieved_all_kernels_gbs": round(alg_bytes / (total_ms * 1e-3) / 1e9, 3),
           "stage_ms": {k: round(v, 2) for k, v in last.items()}, "traffic": None}
    if with_pmc:
        try:   # the dominant KERNEL by rocprofv3 (the stage clock above is host-side wall time around groups of kernels)
            ks = rocprof_kernel_stats([sys.executable, "-c", code], env=env)
            calls = 2     # the worker's warm-up call + its one timed call = two windows
            name, n, tot_ns, avg_ns = ks[0]
            per_window_ms = tot_ns / calls / 1e6
            out.update({"kernel": _short_kernel(name), "kernel_ms": round(per_window_ms, 3), "kernel_launches_per_window": n / calls,
                        "kernel_avg_launch_us": round(avg_ns / 1e3, 1), "achieved": round(alg_bytes / (per_window_ms * 1e-3) / 1e9, 3),
                        "frac": round(alg_bytes / (per_window_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                        "kernels_ms_per_window": round(sum(r[2] for r in ks) / calls / 1e6, 2),
                        "top_kernels_ms_per_window": {_short_kernel(r[0])[:60]: round(r[2] / calls / 1e6, 3) for r in ks[:6]},
                        "achieved_all_kernels_gbs": round(alg_bytes / (sum(r[2] for r in ks) / calls * 1e-9) / 1e9, 3)})
        except Exception as e:
            out["kernel_stats_error"] = repr(e)
        try:
            tot = {}
            for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                rows = rocprof_counter([sys.executable, "-c", code], ctr, env=env)
                tot[ctr] = sum(a * n for a, n in rows.values()) * 1024.0
            calls = 2     # the worker's warm-up call + its one timed call
            out["traffic"] = int((2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / calls)
            out["traffic_detail"] = "all kernels of one window: (2*FETCH_SIZE + WRITE_SIZE) summed over every dispatch of %d calls / %d" % (calls, calls)
        except Exception as e:
            out["traffic_detail"] = "pmc pass failed: %r" % (e,)
    return out


def lgs_leg(rank, local_rank, workers, contig_mb, calls, with_ref, with_pmc):
    """Long-read path (BASELINE configs[3] shape: 20x ONT-like reads, lib/nextpolish2.so ctg_cns_core, BAM -> consensus):
    `workers` worker processes share this rank's GPU (the reference's -p model), each polishes its contig `calls`
    times after a warm-up; rate = polished bp of all workers / span from the common start to the last end."""
    from nextpolish_amd import _native as nat
    d = tempfile.mkdtemp(prefix="np2bench_r%d_" % rank)
    L = int(contig_mb * 1e6)
    st = nat.Stream.synth_long([L], depth=20.0, seed=9000 + rank)
    alg_bytes = st.algorithmic_bytes(False) - L + (L + 3) // 4 + L    # records + 2-bit draft + polished string (path B)
    fa, bam, fofn = os.path.join(d, "g.fa"), os.path.join(d, "r.bam"), os.path.join(d, "bam.fofn")
    st.write_files(fa, bam)
    st.close()
    with open(fofn, "w") as f:
        f.write(bam + "\n")
    go = os.path.join(d, "go")
    env = dict(os.environ, NP2_DEVICE=str(local_rank))
    wt = os.environ.get("NP2_WORKER_THREADS", "2")   # many workers share the host cores of one GPU (round-3 sweep: 24 x 2 > 16 x 4 > 12 x 4)
    env["NP_HOST_THREADS"] = wt
    env["NP_IO_THREADS"] = wt
    lib2 = os.path.join(ROOT, "nextpolish_amd", "lib", "nextpolish2.so")
    ps = []
    for w in range(workers):
        code = LGS_WORKER % dict(root=ROOT, lib=lib2, fa=fa, fofn=fofn, ready=os.path.join(d, "ready%d" % w), go=go, calls=calls)
        ps.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    t_wait = time.time()
    while not all(os.path.exists(os.path.join(d, "ready%d" % w)) for w in range(workers)):
        if any(p.poll() not in (None, 0) for p in ps) or time.time() - t_wait > 300:
            break
        time.sleep(0.01)
    open(go, "w").close()
    outs = [p.communicate() for p in ps]
    bad = [o[1][-300:] for p, o in zip(ps, outs) if p.returncode != 0]
    if bad:
        shutil.rmtree(d, ignore_errors=True)
        return {"error": bad[0]}
    rows = [[float(x) for x in o[0].strip().splitlines()[-1].split()] for o in outs]
    t0, t1 = min(r[0] for r in rows), max(r[1] for r in rows)
    bp = sum(r[2] for r in rows)
    res = {"bp": bp, "seconds": t1 - t0, "workers": workers, "calls_per_worker": calls,
           "s_per_call": round(sum(r[1] - r[0] for r in rows) / (workers * calls), 4),
           "cpu_s_per_mbp": round(sum(r[3] for r in rows) / (bp / 1e6), 4)}
    if rank == 0:
        one = LGS_WORKER % dict(root=ROOT, lib=lib2, fa=fa, fofn=fofn, ready=os.path.join(d, "ready_rf"), go=go, calls=1)
        res["roofline"] = lgs_roofline(one, dict(env, NP_HOST_THREADS="8", NP_IO_THREADS="8"), alg_bytes, with_pmc)
    ref_so = os.path.join(ROOT, "oracle", "_ref", "nextpolish2.so")
    if with_ref and os.path.exists(ref_so):   # the compiled reference on contigs of the same shape: all cores (one process each), one core
        ncore = host_cores()
        files = []

        def make(k):
            s1 = nat.Stream.synth_long([2000000], depth=20.0, seed=4242 + k)
            fa1, bam1, fofn1 = os.path.join(d, "g1_%d.fa" % k), os.path.join(d, "r1_%d.bam" % k), os.path.join(d, "bam1_%d.fofn" % k)
            s1.write_files(fa1, bam1)
            s1.close()
            with open(fofn1, "w") as f:
                f.write(bam1 + "\n")
            return fa1, fofn1
        with ThreadPoolExecutor(min(8, ncore)) as ex:
            files = list(ex.map(make, range(ncore)))

        def run_ref(sel):
            pp = []
            for k in sel:
                code = LGS_WORKER % dict(root=ROOT, lib=ref_so, fa=files[k][0], fofn=files[k][1], ready=os.path.join(d, "readyref%d" % k), go=go, calls=1)
                code = code.replace("once()                                    # warm-up", "pass  #")
                pp.append(subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True))
            rr = []
            for p in pp:
                o, _ = p.communicate()
                if p.returncode != 0:
                    return None
                rr.append([float(x) for x in o.strip().splitlines()[-1].split()])
            return sum(r[2] for r in rr) / 1e6 / (max(r[1] for r in rr) - min(r[0] for r in rr)), max(r[1] for r in rr) - min(r[0] for r in rr)
        allc, one = run_ref(range(ncore)), run_ref([0])
        if allc and one:
            res["cpu_baseline"] = {"value": round(allc[0], 4), "unit": "Mbp/s", "cores": ncore, "kind": "reference", "one_core": round(one[0], 4),
                                   "sample": "%d processes x 2 Mb synthetic contig, 20x ONT-like reads, ctg_cns_core of oracle/_ref/nextpolish2.so, %.1f s; "
                                             "one process alone %.1f s" % (ncore, allc[1], one[1])}
    shutil.rmtree(d, ignore_errors=True)
    return res


def phase_leg(device_index, mb, passes, procs, with_ref):
    """snp_phase (task 3, lib/nextpolish1.so np1_batch_snp_phase): one diploid draft, 30x short read pairs + 20x long reads, both
    record batches resident in HBM; a pass = sites, low-depth correction, links, chain, emit.  Beside it the compiled reference
    (`nextpolish1 snpphase`, BAM files -> FASTA) on all cores and on one."""
    from nextpolish_amd import _native as nat
    from nextpolish_amd import device
    n_ctg = max(1, int(mb // 4))
    lens = [int(mb * 1e6 / n_ctg)] * n_ctg
    t0 = time.time()
    sr, lr = nat.Stream.synth_diploid(lens, seed=9090, sr_holes=2)
    t_gen = time.time() - t0
    bp = int(sr.ctg_len.sum())
    cfg = nat.default_config()
    cfg.read_tlen, cfg.read_len = 2000, 150      # what config_init derives from this short-read BAM (mean insert 400 x 5)
    out = {"metric": "polished Mbp/s (snp_phase, 30x short + 20x long reads, record batches resident in HBM)", "unit": "Mbp/s"}
    ctx = device.Context(device_index)
    try:
        b, bl = ctx.upload(sr), ctx.upload(lr)
        b.snp_phase(bl, cfg)
        times = []
        for _ in range(passes):
            t0 = time.perf_counter()
            b.snp_phase(bl, cfg)
            nat.lib().np1_batch_sync(b.handle)
            times.append(time.perf_counter() - t0)
        alg = sr.algorithmic_bytes(True) + lr.algorithmic_bytes(True)
        best = min(times)
        out.update({"value": round(bp / 1e6 / best, 2), "ms_per_pass": round(best * 1e3, 2), "passes": passes,
                    "achieved_whole_pass_gbs": round(alg / best / 1e9, 2),
                    "roofline": {"bound": "hbm", "kernel": "whole pass (the task's 25 kernels + its host steps; per-kernel times: profiles/r4_snp_phase_kernel_stats_20mb.txt)",
                                 "achieved": round(alg / best / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / best / 1e9 / HBM_PEAK_GBS, 5),
                                 "algorithmic_bytes_per_pass": int(alg), "traffic": None,
                                 "note": "latency- and atomic-bound irregular work (a lane per record, site or region), not a streaming kernel: the fraction says how far"},
                    "config": {"workload": "%.1f Mb diploid draft in %d contigs (0.2 %% heterozygous substitutions, 0.03 %% indel alleles, 0.2 %% draft errors) + 30x PE150 "
                                           "(%d records) + 20x long reads of 8 kb, 4 %% errors (%d records); synthetic, generated in %.1f s"
                                           % (bp / 1e6, n_ctg, sr.n_reads, lr.n_reads, t_gen)}})
        bl.close()
        b.close()
    finally:
        ctx.close()
    # the same draft from files: FASTA + two sorted BAM files in the page cache -> polished contigs; both files go through the
    # device-side ingest, contigs in batches of 4 Mb on two to four lanes (qualities as the generator made them, not binned: the
    # resident measurement above and this one polish the same records)
    td = tempfile.mkdtemp(prefix="np1phase_e2e_")
    try:
        fa, s_bam, l_bam = os.path.join(td, "g.fa"), os.path.join(td, "sr.bam"), os.path.join(td, "lr.bam")
        sr.write_files(fa, s_bam, 1)
        lr.write_files(os.path.join(td, "l.fa"), l_bam, 1)
        exe = os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1")
        batch_bp = 4000000 + 1000
        env = dict(os.environ, NP1_DEVICE=str(device_index), NP_IO_THREADS=str(procs), NP1_BATCH_BP=str(batch_bp))
        t0 = time.time()
        q = subprocess.run([exe, "snpphase", fa, s_bam, l_bam], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
        dt = time.time() - t0
        if q.returncode != 0:
            out["e2e_from_files"] = {"error": q.stderr.decode()[-300:]}
        else:
            e2e = {"mbp_s": None, "cold_process_mbp_s": round(bp / 1e6 / dt, 2), "cold_process_seconds": round(dt, 2),
                   "bam_mb": round((os.path.getsize(s_bam) + os.path.getsize(l_bam)) / 1e6, 1)}
            by_lanes = {}
            n_out = [0]
            for lanes in (2, 3, 4):
                pipe = device.Pipe(device_index, lanes=lanes)
                try:
                    warm = []
                    for _ in range(3):
                        n_out[0] = 0
                        t0 = time.perf_counter()
                        pipe.run_phase_files(fa, s_bam, l_bam, batch_bp=batch_bp, cfg=cfg, sink=lambda name, seq: n_out.__setitem__(0, n_out[0] + len(seq)))
                        warm.append(time.perf_counter() - t0)
                    by_lanes[lanes] = min(warm)
                finally:
                    pipe.close()
            best = min(by_lanes, key=by_lanes.get)
            e2e.update({"mbp_s": round(bp / 1e6 / by_lanes[best], 2), "seconds": round(by_lanes[best], 3), "lanes": best,
                        "mbp_s_by_lanes": {str(k): round(bp / 1e6 / v, 2) for k, v in by_lanes.items()}, "polished_bases_out": n_out[0],
                        "what": "FASTA + short-read BAM + long-read BAM in the page cache (qualities as generated: uniformly random per base, the least compressible case) -> "
                                "polished contigs at a sink: np1_pipe_run_phase_files in a warm process (best of 3 per lane count), both files through the device-side "
                                "ingest, 4 Mb batches; cold_process = the `nextpolish1 snpphase` CLI started from nothing on the same files (HIP start-up and the "
                                "first allocations included, FASTA written, 3 lanes)"})
            out["e2e_from_files"] = e2e
    finally:
        shutil.rmtree(td, ignore_errors=True)
    sr.close()
    lr.close()
    ref = os.path.join(ROOT, "oracle", "_ref", "nextpolish1")
    if with_ref and os.path.exists(ref):
        td = tempfile.mkdtemp(prefix="np1phase_")
        try:
            Ls = 1000000

            def make(k):
                a, c = nat.Stream.synth_diploid([Ls], seed=777 + k, sr_holes=2, prefix="s%dctg" % k)
                fa, s_bam, l_bam = os.path.join(td, "s%d.fa" % k), os.path.join(td, "s%d.sr.bam" % k), os.path.join(td, "s%d.lr.bam" % k)
                a.write_files(fa, s_bam, 1)
                c.write_files(os.path.join(td, "l%d.fa" % k), l_bam, 1)
                return fa, s_bam, l_bam
            with ThreadPoolExecutor(min(8, procs)) as ex:
                files = list(ex.map(make, range(procs)))
            t0 = time.time()
            ps = [subprocess.Popen([ref, "snpphase", fa, s_bam, l_bam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for fa, s_bam, l_bam in files]
            for q in ps:
                q.wait()
            dt_all = time.time() - t0
            t0 = time.time()
            subprocess.run([ref, "snpphase"] + list(files[0]), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            dt_one = time.time() - t0
            out["cpu_baseline"] = {"value": round(procs * Ls / 1e6 / dt_all, 3), "unit": "Mbp/s", "cores": procs, "kind": "reference",
                                   "one_core": round(Ls / 1e6 / dt_one, 3),
                                   "sample": "%d processes x 1 Mb diploid contig of the same kind, nextpolish1 snpphase, two BAM files -> FASTA, %.1f s; one process alone %.1f s"
                                             % (procs, dt_all, dt_one)}
        finally:
            shutil.rmtree(td, ignore_errors=True)
    return out


def _write_files(streams, fa, bam, qual_model):
    from nextpolish_amd import _native as nat
    import ctypes as C
    arr = (C.c_void_p * len(streams))(*[s.handle for s in streams])
    L = nat.lib()
    L.np1_streams_write_files_q.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    t0 = time.time()
    if L.np1_streams_write_files_q(arr, len(streams), fa.encode(), bam.encode(), 1, qual_model) != 0:
        raise RuntimeError(nat.last_error())
    return time.time() - t0


def _time_files(fa, bam, draft_bp, threads, n_records, cold_runs, warm_runs, pipe, expect=None):
    """cold: the CLI, a new process each time; warm: np1_pipe_run_files of this (running) process.  expect = {contig name: md5 of the oracle's
    polished string}: the untimed first warm pass hashes those contigs as they reach the sink and the result carries the comparison."""
    import hashlib
    exe = os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1")
    env = dict(os.environ, NP_IO_THREADS=str(threads), NP_HOST_THREADS=str(threads))
    best, nbytes = 1e9, 0
    for _ in range(cold_runs):
        t0 = time.time()
        q = subprocess.Popen([exe, "scorechain", fa, bam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env)
        n = 0
        for chunk in iter(lambda: q.stdout.read(1 << 24), b""):     # the FASTA text is counted, not kept (3 GB at full size)
            n += len(chunk)
        if q.wait() != 0:
            raise RuntimeError("nextpolish1 scorechain failed")
        best = min(best, time.time() - t0)
        nbytes = n
    import ctypes as C
    from nextpolish_amd import _native as nat
    Ls = nat.lib()
    Ls.np1_pipe_ingest_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    Ls.np1_pipe_ingest_stats.restype = None
    stats = (C.c_double * 5)()
    warm, nout = 1e9, 0
    if pipe is None:      # (cold runs only)
        return {"mbp_s": round(draft_bp / 1e6 / best, 2), "seconds": round(best, 3)}
    seen = {}

    def hashing_sink(name, ptr, n):      # (first pass only: md5 of the contigs the oracle was run on)
        got[0] += n
        nm = name.decode()
        if nm in expect:
            seen[nm] = hashlib.md5(C.string_at(ptr, n)).hexdigest()

    for i in range(warm_runs + 1):
        got = [0]
        t0 = time.time()
        sink = hashing_sink if (i == 0 and expect) else (lambda name, ptr, n: got.__setitem__(0, got[0] + n))
        pipe.run_files(fa, bam, batch_bp=int(os.environ.get("NP1_BATCH_BP", "16000000")), raw_sink=sink)
        if i > 0:                            # the first pass grows the buffers to their final size
            warm = min(warm, time.time() - t0)
        else:
            Ls.np1_pipe_ingest_stats(pipe.handle, stats, 1)      # (the first pass is not part of the decoder's figures either)
        nout = got[0]
    Ls.np1_pipe_ingest_stats(pipe.handle, stats, 1)
    size = os.path.getsize(bam)
    r = {"mbp_s": round(draft_bp / 1e6 / best, 2), "seconds": round(best, 3), "warm_mbp_s": round(draft_bp / 1e6 / warm, 2), "warm_seconds": round(warm, 3),
         "bam_mb": round(size / 1e6, 1), "bam_bytes_per_record": round(size / max(1, n_records), 2), "fasta_bytes_out": nbytes, "bases_out_warm": nout}
    if expect:
        differing = sorted(n for n in expect if seen.get(n) != expect[n])
        r["parity"] = {"contigs": len(expect), "identical": not differing, "differing": differing,
                       "what": "md5 of the contigs the oracle was run on (the streamed passes' parity set), as they left np1_pipe_run_files in the first warm pass: "
                               "BAM + FASTA on disk -> device inflate + CRC + record split + kernels"}
    if stats[4] > 0 and stats[0] > 0:
        # the dominant kernel of the from-files leg: the BGZF block decoder.  Algorithmic bytes of one launch = the compressed
        # bytes it reads + the inflated bytes it writes; time = HIP events around the launch on the lane's stream (np1_ingest.hip), warm passes only
        algo = (stats[2] + stats[3]) / stats[4]
        ms = stats[0] / stats[4]
        r["roofline"] = {"bound": "hbm", "kernel": "BGZF block decoder, a lane per block (k_inflate_lds / k_inflate_lanes: NP1_INFLATE)", "achieved": round(algo / ms / 1e6, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(algo / ms / 1e6 / 8000.0, 6), "traffic": None, "algorithmic_bytes_per_launch": int(algo), "kernel_ms": round(ms, 3),
                         "launches_averaged": int(stats[4]), "inflated_out_gbs": round(stats[3] / stats[0] / 1e6, 2), "crc_ms": round(stats[1] / stats[4], 3),
                         "decoder_ms_per_pass": round(stats[0] / max(1, warm_runs), 1),
                         "what": "compressed bytes in + inflated bytes out of one launch / its HIP-event time; launches of the lanes overlap, so decoder_ms_per_pass can exceed the pass"}
    return r


def e2e_from_files(streams, draft_bp, threads, n_records, expect=None):
    """Scope 2: one FASTA + one sorted BAM on disk (page cache) -> polished FASTA.  Full size on a BAM with Illumina-like binned
    qualities (cold CLI process: HIP start-up, first-touch allocations, BGZF inflate + record split on the device, H2D, kernels,
    D2H, FASTA text; warm: the same files through np1_pipe_run_files of a running process); the first batch alone on a BAM
    without qualities and on one with uniformly random qualities, for the two ends of the compressibility range."""
    from nextpolish_amd.device import Pipe
    d = tempfile.mkdtemp(prefix="np1e2e_")
    out = {}
    pipe = Pipe(int(os.environ.get("LOCAL_RANK", "0")), lanes=int(os.environ.get("NP1_E2E_LANES", "3")))      # round-4 sweep (tests/tools/r4_e2e_lanes.py): 3 > 2 > 4
    try:
        fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
        t_write = _write_files(streams, fa, bam, 1)
        full = _time_files(fa, bam, draft_bp, threads, n_records, 2, 2, pipe, expect=expect)
        full["qualities"] = "Illumina-like, binned (2/12/23/37; ~93 % in the top bin), BGZF level 1"
        full["write_seconds"] = round(t_write, 1)
        bt = max(1, threads // 8)      # what a rank gets of this box's cores when eight ranks share them
        b8 = _time_files(fa, bam, draft_bp, bt, n_records, 1, 0, None)
        full["at_8_rank_budget"] = {"host_threads": bt, "mbp_s": b8["mbp_s"], "seconds": b8["seconds"], "fraction_of_full_budget": round(b8["mbp_s"] / max(1e-9, full["mbp_s"]), 3),
                                    "what": "the cold CLI run again with NP_IO_THREADS = this box's cores / 8"}
        full["what"] = ("cold: nextpolish1 scorechain g.fa r.bam > out.fa, new process each time (HIP start-up and first-touch allocations inside), "
                        "best of 2, files in the page cache, %d host threads; warm: the same files through np1_pipe_run_files of a running "
                        "process (device BGZF inflate + CRC + record split + kernels + D2H), best of 2 after one pass" % threads)
        out.update(full)
        os.remove(bam)
        one = streams[:1]
        bp1 = int(sum(int(x) for x in one[0].ctg_len))
        for key, qm, label in (("no_qualities", 0, "no base qualities (0xff): ~9:1"), ("random_qualities", 2, "uniformly random qualities in [25, 40]: incompressible")):
            _write_files(one, fa, bam, qm)
            r = _time_files(fa, bam, bp1, threads, one[0].n_reads, 1, 2, pipe)
            r["qualities"] = label
            r["slice"] = "first batch only: %.1f Mb, %d records" % (bp1 / 1e6, one[0].n_reads)
            out[key] = r
            os.remove(bam)
        return out
    except Exception as e:     # the headline line must survive a failure of this leg
        out["error"] = repr(e)
        return out
    finally:
        pipe.close()
        shutil.rmtree(d, ignore_errors=True)


def parity_check(pipe, streams, budget_bp, per_batch_cap_bp=40000000, big_seconds=180.0):
    """Polished strings of this run (the last streamed pass) against the CPU oracle: the shortest contigs of the draft up to budget_bp AND
    the shortest contig of EVERY batch (so that each batch of the pass is represented; a batch whose shortest contig exceeds
    per_batch_cap_bp is listed as unchecked).  The oracle calls run side by side on the host cores."""
    import ctypes as C
    import hashlib
    from nextpolish_amd import _native as nat
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    cand = sorted((int(st.ctg_len[c]), k, c) for k, st in enumerate(streams) for c in range(st.n_contigs))
    chosen, bp = [], 0
    for L, k, c in cand:
        if chosen and bp + L > budget_bp:
            break
        chosen.append((L, k, c))
        bp += L
    unchecked = []
    for k, st in enumerate(streams):
        if any(kk == k for _, kk, _ in chosen) or st.n_contigs == 0:
            continue
        L, c = min((int(st.ctg_len[c]), c) for c in range(st.n_contigs))
        if L <= per_batch_cap_bp or L / 1.9e6 <= big_seconds:      # (the oracle walks ~2 Mbp/s on one core; the contigs are checked side by side)
            chosen.append((L, k, c))
        else:
            unchecked.append(k)
    t0 = time.time()

    def one(item):
        L, k, c = item
        ln = C.c_int64(0)
        p = nat.lib().np1_pipe_result(pipe.handle, k, c, C.byref(ln))
        got = C.string_at(p, ln.value)
        want = hashlib.md5(ob.score_chain(streams[k], c).encode()).hexdigest()
        oracle_md5[streams[k].names[c]] = want
        return None if hashlib.md5(got).hexdigest() == want else streams[k].names[c]
    oracle_md5 = {}      # name -> md5 of the oracle's string: the from-files leg compares its own output of the same contigs with these
    with ThreadPoolExecutor(max(1, min(host_cores(), len(chosen)))) as ex:
        bad = [x for x in ex.map(one, sorted(chosen, reverse=True)) if x is not None]
    return {"oracle_md5": oracle_md5, "contigs": len(chosen), "draft_bp": sum(x[0] for x in chosen), "batches_represented": len({k for _, k, _ in chosen}), "batches": len(streams),
            "batches_unchecked": unchecked, "identical": not bad, "differing": bad, "oracle_seconds": round(time.time() - t0, 1),
            "what": "md5 of the polished strings of the timed streamed passes vs oracle/np1_oracle (CPU restatement): the shortest contigs of the draft and "
                    "the shortest contig of every batch (up to %d Mb each, or whatever the oracle walks in %d s on one core: VERDICT r4 weak 3)" % (per_batch_cap_bp // 1000000, int(big_seconds))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--streamed-passes", type=int, default=3, help="timed passes of the PCIe-inclusive sub-measurement (inputs in pinned host memory)")
    ap.add_argument("--parity-mb", type=float, default=8.0, help="draft bases compared with the CPU oracle inside the run")
    ap.add_argument("--parity-big-seconds", type=float, default=180.0, help="a batch whose shortest contig is longer than 40 Mb is still checked when the oracle (one core, ~2 Mbp/s) walks that contig within this many seconds")
    ap.add_argument("--workload", default="c5_3gb_30x", choices=sorted(WORKLOADS))
    ap.add_argument("--lanes", type=int, default=2, help="batches in flight on the device")
    ap.add_argument("--batch-mb", type=float, default=0.0, help="draft bases per batch (Mb); 0 = the workload's own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mb", type=float, default=3.0)
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes (roofline.traffic = null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--pmc-batches", type=int, default=4, help="batches of the workload the two rocprofv3 --pmc child runs polish (generating all 600 M records twice more is most of a full-size run)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the from-files leg")
    ap.add_argument("--no-lgs", action="store_true", help="skip the long-read (nextpolish2) leg")
    ap.add_argument("--lgs-workers", type=int, default=24, help="worker processes per GPU of the long-read leg (x NP2_WORKER_THREADS host threads each, default 2)")
    ap.add_argument("--lgs-mb", type=float, default=5.0, help="contig length (Mb) each long-read worker polishes")
    ap.add_argument("--lgs-calls", type=int, default=24, help="timed calls per long-read worker (the rate is bp over the span from the common start to the last end; measured: 0.63 s of ramp and stragglers + 0.70 s per round of calls, so 4 / 8 calls read 140 / 154 Mbp/s of a steady state of 171: a worker of a real run polishes hundreds of windows)")
    ap.add_argument("--no-lgs-config4", action="store_true", help="skip the 100 Mb / 67-contig long-read run (BASELINE configs[3] at its stated size)")
    ap.add_argument("--no-phase", action="store_true", help="skip the snp_phase (task 3) leg")
    ap.add_argument("--phase-mb", type=float, default=20.0, help="draft length (Mb) of the snp_phase leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # host threads of this rank = the box's cores / ranks (generator, BAM writer, loaders; read once when the library loads)
    per_rank = max(1, host_cores() // max(1, world))
    os.environ.setdefault("NP_HOST_THREADS", str(per_rank))
    os.environ.setdefault("NP_IO_THREADS", str(per_rank))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import ctypes as C
    from nextpolish_amd import _native as nat
    from nextpolish_amd.device import Pipe
    from nextpolish_amd.nextpolish1 import plan_batches
    from nextpolish_amd.shard import deal_contigs

    total, depth, lo, hi, batch_bp = WORKLOADS[args.workload]
    if args.batch_mb > 0:
        batch_bp = int(args.batch_mb * 1e6)
    lens = contig_lengths(total, lo, hi)
    names = ["c%04d" % i for i in range(len(lens))]
    lmap = dict(zip(names, lens))
    owner = deal_contigs(names, lmap, world)                      # the fixed draft is split over the ranks (strong scaling)
    mine = [n for n in names if owner[n] == rank]
    batches = plan_batches(mine, lmap, batch_bp)
    if args.pmc_child:
        batches = batches[:max(1, args.pmc_batches)]
    blens = [[lmap[n] for n in b] for b in batches]
    draft_bp_total = sum(lens)
    my_bp = sum(lmap[n] for n in mine)
    ncpu = host_cores()
    t_gen = time.time()
    # chromosome-sized contigs are generated segment by segment on all of the rank's threads (np_synth.cpp): few streams at a time
    outer = 2 if max(lens) >= (16 << 20) else max(1, min(per_rank, len(blens)))
    with ThreadPoolExecutor(outer) as ex:
        streams = list(ex.map(lambda k: nat.Stream.synth(blens[k], depth=depth, seed=20250117 + 1000 * int(batches[k][0][1:]),
                                                         prefix="b%dc" % int(batches[k][0][1:])), range(len(blens))))
    t_gen = time.time() - t_gen
    n_reads = sum(s.n_reads for s in streams)
    alg_in = [s.algorithmic_bytes(False) for s in streams]   # records (32 + 4 n_cigar + ceil(l/2)) + draft, per batch
    # The upload forms (2-bit bases, compact record fields: DESIGN.md section 4) are built ONCE per stream on the host, outside the timed
    # steps; a stream that is polished once pays for them once.  Timed here and reported next to `value` (upload_forms_build_s,
    # streamed_single_use) so that the line says what its scope leaves out.
    L0 = nat.lib()
    L0.np1_stream_upload_bytes.restype = C.c_uint64
    L0.np1_stream_upload_bytes.argtypes = [C.c_void_p]
    t_forms = time.time()
    with ThreadPoolExecutor(max(1, min(per_rank, len(streams)))) as ex:
        list(ex.map(lambda s: L0.np1_stream_upload_bytes(s.handle), streams))     # builds the forms (np1_device.hip: stream_facts)
    t_forms = time.time() - t_forms
    t_pin = time.time()
    for s in streams:
        s.pin()
    t_pin = time.time() - t_pin
    pipe = Pipe(local_rank, lanes=args.lanes)
    cfg = nat.default_config()
    L = nat.lib()
    L.np1_pipe_upload.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int]
    L.np1_pipe_run_resident.argtypes = [C.c_void_p, C.POINTER(nat.Configure), C.c_int, C.c_int]
    L.np1_pipe_resident_batch.argtypes = [C.c_void_p, C.c_int]
    L.np1_pipe_resident_batch.restype = C.c_void_p
    L.np1_pipe_run_resident_timed.argtypes = [C.c_void_p, C.c_int, C.POINTER(nat.Configure), C.POINTER(C.c_float)]
    harr = (C.c_void_p * len(streams))(*[s.handle for s in streams])

    def resident(passes):
        if passes > 0 and L.np1_pipe_run_resident(pipe.handle, C.byref(cfg), 1, passes) != 0:
            raise SystemExit("run_resident: " + nat.last_error())

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    if args.pmc_child:   # short profiled run for pmc_traffic(): resident passes only, no JSON, no baseline
        if L.np1_pipe_upload(pipe.handle, harr, len(streams)) != 0:
            raise SystemExit("upload: " + nat.last_error())
        resident(args.warmup + args.steps)
        pipe.close()
        return

    # ---- PCIe-inclusive passes (reported beside `value`, never as it): pinned host -> H2D -> kernels -> D2H, double-buffered on the lanes
    pipe.run(streams, cfg=cfg, fetch=False)              # warm: lane batches grow to their final size
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.streamed_passes):
        pipe.run(streams, cfg=cfg, fetch=False)
    sync_all()
    dt_str = max_over_ranks(time.perf_counter() - t0)
    L.np1_stream_upload_bytes.restype = C.c_uint64
    L.np1_stream_upload_bytes.argtypes = [C.c_void_p]
    h2d_bytes = sum(int(L.np1_stream_upload_bytes(s.handle)) for s in streams)   # what really crosses PCIe per pass (2-bit bases, 16-bit counts, no offsets)
    parity = parity_check(pipe, streams, int(args.parity_mb * 1e6), big_seconds=args.parity_big_seconds) if rank == 0 else None
    streamed_lengths = pipe.result_lengths(streams)

    # ---- the timed steps (SURVEY 8d scope 1): every batch of the rank's share of the draft resident in HBM when the clock starts; a step = one
    # score_chain pass over all of them on the lanes (no H2D / D2H inside).  W untimed steps, then exactly K between two barriers.
    if L.np1_pipe_upload(pipe.handle, harr, len(streams)) != 0:
        raise SystemExit("upload: " + nat.last_error())
    resident(args.warmup)
    sync_all()
    t0 = time.perf_counter()
    resident(args.steps)
    sync_all()
    dt = max_over_ranks(time.perf_counter() - t0)

    # ---- per-stage HIP-event timing on the pipeline's own stream (separate instrumented passes, one lane, batch by batch)
    stage_acc, launches, polished, updates = {}, 0, [], 0
    n_inst = 2
    for k in range(len(streams)):
        b = L.np1_pipe_resident_batch(pipe.handle, k)
        for _ in range(n_inst):
            ms = (C.c_float * nat.NP1_MAX_STAGES)()
            if L.np1_pipe_run_resident_timed(pipe.handle, k, C.byref(cfg), ms) != 0:
                raise SystemExit("timed pass: " + nat.last_error())
            for i in range(L.np1_stage_count()):
                nm = L.np1_stage_name(i).decode()
                stage_acc[nm] = stage_acc.get(nm, 0.0) + float(ms[i])
            launches += 1
        polished.append(sum(int(L.np1_batch_result_len(b, c)) for c in range(streams[k].n_contigs)))
        updates += int(L.np1_batch_update_count(b))
    stage_ms = {k: v / launches for k, v in stage_acc.items()}          # average per launch (= per batch)
    dom = max(stage_ms, key=lambda k: stage_ms[k])
    alg_per_launch = (sum(alg_in) + sum(polished)) / float(len(streams))
    achieved = alg_per_launch / (stage_ms[dom] * 1e-3) / 1e9
    assert streamed_lengths == sum(polished)

    traffic, traffic_note = None, "not collected"
    if rank == 0 and world == 1 and not args.no_pmc:
        sub = {"tile": "k_tile3", "vote": "k_vote", "rows": "k_rows"}.get(dom, dom)
        traffic, traffic_note = pmc_traffic(args, sub)
        if traffic:
            nb = min(len(streams), max(1, args.pmc_batches))
            traffic["algorithmic_bytes_per_launch_of_the_profiled_batches"] = int((sum(alg_in[:nb]) + sum(polished[:nb])) / nb)
            traffic["ratio_to_algorithmic"] = round(traffic["bytes"] / traffic["algorithmic_bytes_per_launch_of_the_profiled_batches"], 3)
    e2e = None
    if rank == 0 and world == 1 and not args.no_e2e:
        pipe.close()
        pipe = None
        e2e = e2e_from_files(streams, draft_bp_total, ncpu, n_reads, expect=(parity or {}).get("oracle_md5"))
    if pipe is not None:
        pipe.close()
    lgs = None
    if not args.no_lgs:
        if world > 1:
            dist.barrier()
        lgs_workers = args.lgs_workers if world == 1 else max(2, min(args.lgs_workers, (ncpu * 3 // 4) // world))   # the ranks share the host cores
        lgs = lgs_leg(rank, local_rank, lgs_workers, args.lgs_mb, args.lgs_calls,
                      rank == 0 and world == 1 and not args.no_cpu_baseline, rank == 0 and world == 1 and not args.no_pmc)
        if world == 1 and rank == 0 and "error" not in lgs:
            # the same leg on the host budget of one rank of eight (this box's cores / 8): does the feed of one GPU still hold?
            bt = max(2, ncpu // 8)
            wt = int(os.environ.get("NP2_WORKER_THREADS", "2"))
            l8 = lgs_leg(1, local_rank, max(1, bt // wt), args.lgs_mb, max(2, args.lgs_calls // 4), False, False)
            if "error" not in l8:
                v8 = l8["bp"] / 1e6 / l8["seconds"]
                lgs["at_8_rank_budget"] = {"host_threads": bt, "workers": l8["workers"], "mbp_s": round(v8, 2),
                                           "fraction_of_full_budget": round(v8 / max(1e-9, lgs["bp"] / 1e6 / lgs["seconds"]), 3),
                                           "what": "workers x threads limited to this box's cores / 8, as when eight ranks share the host"}
        if world == 1 and rank == 0 and "error" not in lgs and not args.no_lgs_config4:
            # BASELINE configs[3] at its stated size: ~100 Mb in 67 contigs (four of them two or three windows), 20x ONT-like reads, the contigs
            # in 8 groups (one FASTA + BAM each) taken by worker processes like the reference's -p model: from cold processes to the last
            # consensus, every contig compared with the md5 the compiled reference produced (tests/tools/check_config4.py, tests/golden/)
            env4 = dict(os.environ, NP2_DEVICE=str(local_rank), NP_HOST_THREADS=os.environ.get("NP2_WORKER_THREADS", "2"), NP_IO_THREADS=os.environ.get("NP2_WORKER_THREADS", "2"))
            q = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "check_config4.py"), "--procs", "8"], capture_output=True, text=True, env=env4)
            try:
                c4 = json.loads(q.stdout.strip().splitlines()[-1])
                lgs["config4"] = {"mbp_s": c4.get("mbp_s"), "draft_bp": c4.get("draft_bp"), "contigs": c4.get("contigs"), "groups": c4.get("groups"), "procs": c4.get("procs"),
                                  "polish_s": c4.get("polish_s"), "mismatches_vs_reference_golden": c4.get("mismatches"), "missing": c4.get("missing"),
                                  "what": "100 Mb / 67 contigs / 20x ONT-like (BASELINE configs[3]), 8 cold worker processes sharing the GPU, process start and "
                                          "first allocations included; every contig's md5 against the compiled reference's"}
            except Exception as e:      # noqa: BLE001
                lgs["config4"] = {"error": (q.stderr or str(e))[-300:]}
        if world > 1:   # whole job: bp of all ranks over the slowest rank's span
            tt = torch.tensor([float(lgs.get("bp", 0)), float(lgs.get("seconds", 0)), 1.0 if "error" in lgs else 0.0], device="cuda", dtype=torch.float64)
            bp_sum = tt.clone()
            dist.all_reduce(bp_sum, op=dist.ReduceOp.SUM)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            if bp_sum[2].item() == 0:
                lgs["bp"], lgs["seconds"] = bp_sum[0].item(), tt[1].item()
            else:
                lgs.setdefault("error", "a rank failed")
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = draft_bp_total / 1e6 / (dt / args.steps)
        streamed_v = draft_bp_total / 1e6 / (dt_str / args.streamed_passes)
        out = {
            "metric": "polished Mbp/s (score_chain, 30x short reads, one fixed draft; record batches resident in HBM -> kernels -> polished strings in HBM)",
            "value": round(value, 3), "unit": "Mbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8/int64", "data": "synthetic",
            "config": {"workload": "%s: %.1f Mb synthetic draft in %d contigs (log-uniform %g-%g bp) + %.0fx simulated 2x150 bp PE reads "
                                   "(%d records on rank 0), dealt longest-first over %d GPU(s), %d batches of <= %.0f Mb on rank 0; "
                                   "one step = one score_chain pass over the whole draft, every batch resident in HBM when the timed region starts"
                                   % (args.workload, draft_bp_total / 1e6, len(lens), lo, hi, depth, n_reads, world, len(batches), batch_bp / 1e6),
                       "slot_votes_per_step_rank0": updates // n_inst, "lanes": args.lanes,
                       "parallelism": "contigs dealt longest-first x%d (no collective)" % world,
                       "synth_seconds": round(t_gen, 1), "host_cores": ncpu, "host_threads_per_rank": per_rank,
                       },
            "streamed": {"mbp_s": round(streamed_v, 2), "ms_per_pass": round(dt_str / args.streamed_passes * 1e3, 3), "passes": args.streamed_passes,
                         "h2d_gb_per_s_rank0": round(h2d_bytes * args.streamed_passes / dt_str / 1e9, 2), "h2d_bytes_per_draft_bp": round(h2d_bytes / max(1, sum(int(x.ctg_len.sum()) for x in streams)), 2),
                         "what": "the PCIe-inclusive rate of the same pass (never `value`): decoded records in pinned host memory -> H2D -> kernels -> D2H of the polished strings, "
                                 "double-buffered on %d lanes; bound by the H2D of the record stream" % args.lanes},
            "upload_forms": {"build_s": round(t_forms, 3), "pin_s": round(t_pin, 3), "host_threads": max(1, min(per_rank, len(streams))),
                             "streamed_single_use_mbp_s": round(my_bp / 1e6 / (t_forms + dt_str / args.streamed_passes), 2),
                             "what": "the streamed passes upload the records in forms (2-bit bases, compact record fields) that are built once per stream on the host BEFORE the "
                                     "timed passes (build_s, on host_threads threads; pin_s = page-locking them); streamed_single_use = this rank's draft / (build_s + one "
                                     "streamed pass): the rate of a decoded stream that is polished exactly once.  From files the device-side ingest never builds them (e2e_from_files)"},
            "parity": {k: v for k, v in parity.items() if k != "oracle_md5"} if parity else parity,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic["bytes"] if traffic else None,
                         "traffic_detail": traffic if traffic else traffic_note,
                         "algorithmic_bytes_per_launch": int(alg_per_launch), "kernel_ms": round(stage_ms[dom], 4),
                         "launches_averaged": launches, "stage_ms": {k: round(v, 4) for k, v in stage_ms.items()},
                         "achieved_whole_pass_gbs": round((sum(alg_in) + sum(polished)) / (dt / args.steps) / 1e9, 2)},
        }
        if parity is not None and not parity["identical"] and not os.environ.get("NP1_ABLATE"):   # (NP1_ABLATE: timing experiments with wrong results)
            raise SystemExit("bench: the polished strings differ from the oracle: %r" % (parity,))
        if e2e is not None:
            out["e2e_from_files"] = e2e
            fp = e2e.get("parity")
            if fp is not None and not fp["identical"] and not os.environ.get("NP1_ABLATE"):
                raise SystemExit("bench: the from-files output differs from the oracle: %r" % (fp,))
        if lgs is not None:
            if "error" in lgs:
                out["lgs"] = lgs
            else:
                out["lgs"] = {"metric": "polished Mbp/s (ctg_cns_core, long reads, sorted BAM in the page cache -> consensus, warm workers)",
                              "value": round(lgs["bp"] / 1e6 / lgs["seconds"], 3), "unit": "Mbp/s", "n_gpus": world,
                              "config": {"workload": "%.1f Mb synthetic contig + 20x ONT-like reads (8 kb, 7%% errors) per worker, %d worker processes per GPU, "
                                                     "%d calls each, %s host threads per worker" % (args.lgs_mb, lgs["workers"], args.lgs_calls, os.environ.get("NP2_WORKER_THREADS", "2"))},
                              "s_per_call": lgs["s_per_call"], "host_cpu_s_per_mbp": lgs["cpu_s_per_mbp"]}
                for k in ("roofline", "cpu_baseline", "config4", "at_8_rank_budget"):
                    if k in lgs:
                        out["lgs"][k] = lgs[k]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(depth, args.cpu_sample_mb, ncpu)
        if world == 1 and not args.no_phase:
            try:
                out["snp_phase"] = phase_leg(local_rank, args.phase_mb, 3, ncpu, not args.no_cpu_baseline)
            except Exception as e:   # the headline line must survive a failure of an extra leg
                out["snp_phase"] = {"error": str(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
