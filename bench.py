#!/usr/bin/env python
"""Headline benchmark: short-read score_chain polishing throughput (BASELINE.json metric).

One "step" = one full score_chain pass (every kernel of nextpolish_amd/csrc/np1_device.hip's launch
sequence) over one batch of synthetic draft contigs + position-sorted short reads that is already
resident in HBM.  Default workload = BASELINE.json configs[1]: 5 Mb draft, 50x PE150.
With --gpus N every rank polishes its own 5 Mb shard (contigs are independent units: weak scaling,
no data-path collective; reference: source/lib/nextpolish1.py:181-189,223-224).

Prints ONE JSON line on rank 0 (see the driver contract in the task description), including
  roofline     : achieved algorithmic HBM bytes/s of the dominant kernel (k_vote) vs the 8 TB/s peak
  cpu_baseline : the reference CPU path (oracle/_ref/nextpolish1, or the oracle port) timed on a
                 bounded sample of the same workload on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
WORKLOADS = {
    # name: (contig truth lengths, depth)            -- SURVEY.md §8d synthetic shapes
    "c2_5mb_50x": ([2500000, 1500000, 1000000], 50.0),
    "small_1mb_50x": ([600000, 400000], 50.0),
    "c3shard_12mb_30x": ([3000000, 2500000, 2000000, 1500000, 1200000, 1000000, 800000], 30.0),
}


def cpu_baseline(stream_factory, sample_len, depth, seed):
    """Times the reference C path on a bounded sample (own process, 1 thread)."""
    from nextpolish_amd import _native as nat
    st = stream_factory([sample_len], depth, seed)
    bp = int(st.ctg_len.sum())
    ref = os.path.join(ROOT, "oracle", "_ref", "nextpolish1")
    if os.path.exists(ref):
        with tempfile.TemporaryDirectory() as td:
            fa, bam = os.path.join(td, "s.fa"), os.path.join(td, "s.bam")
            st.write_files(fa, bam, 1)
            t0 = time.time()
            subprocess.run([ref, "scorechain", fa, bam], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            dt = time.time() - t0
        kind = "reference"
    else:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_binding as ob
        t0 = time.time()
        for i in range(st.n_contigs):
            ob.score_chain(st, i)
        dt = time.time() - t0
        kind = "port"
    return {"value": round(bp / 1e6 / dt, 4), "unit": "Mbp/s", "cores": 1, "kind": kind,
            "sample": "%.1f Mb synthetic draft, %.0fx PE150, score_chain, BAM(+BGZF inflate) -> FASTA, %.1f s"
                      % (bp / 1e6, depth, dt)}


def pmc_traffic(args, kernel_substr):
    """HBM bytes of the dominant kernel per launch from rocprofv3 PMC counters: two separate --pmc passes
    (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950) over a short child run of this script.
    Units/corrections per MI355X_MICROARCH.md (HBM section): both counters are KiB per dispatch; on gfx950
    FETCH_SIZE reports half of the bytes of wide (16 B/lane) coalesced reads, which is how this kernel stages its
    inputs, so the read side is doubled; WRITE_SIZE is taken as is."""
    import shutil
    import sqlite3
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not found"
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        td = tempfile.mkdtemp(prefix="np1pmc_", dir="/tmp")
        cmd = [exe, "--pmc", ctr, "-d", td, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
               "--workload", args.workload, "--steps", "2", "--warmup", "1"]
        try:
            subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True,
                           cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
            db = None
            for root, _d, files in os.walk(td):
                for f in files:
                    if f.endswith(".db"):
                        db = os.path.join(root, f)
            c = sqlite3.connect(db)
            row = c.execute("select avg(value) from counters_collection where counter_name=? and kernel_name like ?",
                            (ctr, "%" + kernel_substr + "%")).fetchone()
            vals[ctr] = float(row[0])
        except Exception as e:   # profiling is best effort: the bench line stays valid without it
            return None, "pmc pass failed: %r" % (e,)
        finally:
            shutil.rmtree(td, ignore_errors=True)
    fetch_b, write_b = vals["FETCH_SIZE"] * 1024.0, vals["WRITE_SIZE"] * 1024.0
    return {"bytes": int(2 * fetch_b + write_b), "fetch_size_kib_raw": round(vals["FETCH_SIZE"], 1),
            "write_size_kib_raw": round(vals["WRITE_SIZE"], 1), "correction": "2*FETCH_SIZE + WRITE_SIZE (gfx950)"}, None


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


def lgs_leg(rank, local_rank, workers, contig_mb, calls, with_ref):
    """Long-read path (BASELINE configs[3] shape: 20x ONT-like reads, lib/nextpolish2.so ctg_cns_core, BAM -> consensus):
    `workers` worker processes share this rank's GPU (the reference's -p model), each polishes its contig `calls`
    times after a warm-up; rate = polished bp of all workers / span from the common start to the last end."""
    import shutil, subprocess, tempfile
    from nextpolish_amd import _native as nat
    d = tempfile.mkdtemp(prefix="np2bench_r%d_" % rank)
    L = int(contig_mb * 1e6)
    st = nat.Stream.synth_long([L], depth=20.0, seed=9000 + rank)
    fa, bam, fofn = os.path.join(d, "g.fa"), os.path.join(d, "r.bam"), os.path.join(d, "bam.fofn")
    st.write_files(fa, bam)
    st.close()
    with open(fofn, "w") as f:
        f.write(bam + "\n")
    go = os.path.join(d, "go")
    env = dict(os.environ, NP2_DEVICE=str(local_rank))
    env.setdefault("NP_HOST_THREADS", "4")   # several workers share the host cores of one GPU
    env.setdefault("NP_IO_THREADS", "4")
    ps = []
    for w in range(workers):
        code = LGS_WORKER % dict(root=ROOT, lib=os.path.join(ROOT, "nextpolish_amd", "lib", "nextpolish2.so"), fa=fa, fofn=fofn,
                                 ready=os.path.join(d, "ready%d" % w), go=go, calls=calls)
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
    ref_so = os.path.join(ROOT, "oracle", "_ref", "nextpolish2.so")
    if with_ref and os.path.exists(ref_so):   # the compiled reference, one core, on a 1 Mb contig of the same shape
        st = nat.Stream.synth_long([1000000], depth=20.0, seed=4242)
        fa1, bam1, fofn1 = os.path.join(d, "g1.fa"), os.path.join(d, "r1.bam"), os.path.join(d, "bam1.fofn")
        st.write_files(fa1, bam1)
        st.close()
        with open(fofn1, "w") as f:
            f.write(bam1 + "\n")
        code = LGS_WORKER % dict(root=ROOT, lib=ref_so, fa=fa1, fofn=fofn1, ready=os.path.join(d, "readyref"), go=go, calls=1)
        code = code.replace("once()                                    # warm-up", "pass  #")
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
        if p.returncode == 0:
            r = [float(x) for x in p.stdout.strip().splitlines()[-1].split()]
            res["cpu_baseline"] = {"value": round(r[2] / 1e6 / (r[1] - r[0]), 4), "unit": "Mbp/s", "cores": 1, "kind": "reference",
                                   "sample": "1 Mb synthetic contig, 20x ONT-like reads, ctg_cns_core of oracle/_ref/nextpolish2.so, %.1f s" % (r[1] - r[0])}
    shutil.rmtree(d, ignore_errors=True)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c2_5mb_50x", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-mb", type=float, default=3.0)
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc passes (roofline.traffic = null)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-lgs", action="store_true", help="skip the long-read (nextpolish2) leg")
    ap.add_argument("--lgs-workers", type=int, default=8, help="worker processes per GPU of the long-read leg")
    ap.add_argument("--lgs-mb", type=float, default=5.0, help="contig length (Mb) each long-read worker polishes")
    ap.add_argument("--lgs-calls", type=int, default=4)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from nextpolish_amd import _native as nat
    from nextpolish_amd.device import Context

    lens, depth = WORKLOADS[args.workload]

    def make_stream(contig_lens, d, seed):
        return nat.Stream.synth(contig_lens, depth=d, seed=seed, prefix="r%dctg" % rank)

    # per-rank shard: same shape, different seed (weak scaling)
    st = make_stream(lens, depth, 20250117 + 2 + 1000 * rank)
    draft_bp = int(st.ctg_len.sum())
    alg_bytes = st.algorithmic_bytes(False)   # records (32 + 4 n_cigar + ceil(l/2)) + draft
    ctx = Context(local_rank)
    ctx.upload(st).close()        # first upload pays hipMalloc; time the second one (pageable host memory -> HBM)
    t_up = time.perf_counter()
    batch = ctx.upload(st)
    upload_ms = (time.perf_counter() - t_up) * 1e3
    cfg = nat.default_config()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        batch.score_chain(cfg)
    nat.lib().np1_batch_sync(batch.handle)
    if args.pmc_child:   # short profiled run for pmc_traffic(): no JSON, no baseline
        for _ in range(args.steps):
            batch.score_chain(cfg)
        nat.lib().np1_batch_sync(batch.handle)
        batch.close()
        return
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.score_chain(cfg)
    nat.lib().np1_batch_sync(batch.handle)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    polished_len = sum(len(s) for s in batch.results())
    alg_bytes_total = alg_bytes + polished_len
    updates = batch.update_count()

    # per-stage HIP-event timing on the pipeline's own stream (separate instrumented passes)
    stage_acc = {}
    n_inst = 5
    for _ in range(n_inst):
        ms = batch.score_chain(cfg, timed=True)
        for k, v in ms.items():
            stage_acc[k] = stage_acc.get(k, 0.0) + v / n_inst
    dom = max(stage_acc, key=lambda k: stage_acc[k])
    dom_ms = stage_acc[dom]
    achieved = alg_bytes_total / (dom_ms * 1e-3) / 1e9

    traffic, traffic_note = None, "not collected"
    if rank == 0 and world == 1 and not args.no_pmc:
        sub = {"tile": "k_tile3", "vote": "k_vote", "rows": "k_rows"}.get(dom, dom)
        traffic, traffic_note = pmc_traffic(args, sub)
    lgs = None
    if not args.no_lgs and not args.pmc_child:
        if world > 1:
            dist.barrier()
        lgs = lgs_leg(rank, local_rank, args.lgs_workers, args.lgs_mb, args.lgs_calls, rank == 0 and world == 1 and not args.no_cpu_baseline)
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
        value = world * draft_bp / 1e6 / (dt / args.steps)
        out = {
            "metric": "polished Mbp/s (score_chain, short reads, inputs resident in HBM)",
            "value": round(value, 3), "unit": "Mbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8/int64", "data": "synthetic",
            "config": {"workload": "%s: %.2f Mb synthetic draft (%d contigs) + %.0fx simulated 2x150 bp PE reads per GPU, "
                                   "one score_chain pass" % (args.workload, draft_bp / 1e6, st.n_contigs, depth),
                       "reads_per_gpu": st.n_reads, "slot_votes_per_step": updates,
                       "parallelism": "contig-sharded x%d (no collective)" % world,
                       "h2d_upload_ms_not_in_value": round(upload_ms, 3),
                       "pcie_inclusive_mbp_s": round(draft_bp / 1e6 / ((ms_per_step + upload_ms) * 1e-3), 1)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic["bytes"] if traffic else None,
                         "traffic_detail": traffic if traffic else traffic_note,
                         "algorithmic_bytes_per_launch": alg_bytes_total, "kernel_ms": round(dom_ms, 4),
                         "stage_ms": {k: round(v, 4) for k, v in stage_acc.items()}},
        }
        if lgs is not None:
            if "error" in lgs:
                out["lgs"] = lgs
            else:
                out["lgs"] = {"metric": "polished Mbp/s (ctg_cns_core, long reads, sorted BAM in the page cache -> consensus, warm workers)",
                              "value": round(lgs["bp"] / 1e6 / lgs["seconds"], 3), "unit": "Mbp/s", "n_gpus": world,
                              "config": {"workload": "%.1f Mb synthetic contig + 20x ONT-like reads (8 kb, 7%% errors) per worker, %d worker processes per GPU, "
                                                     "%d calls each, %s host threads per worker" % (args.lgs_mb, args.lgs_workers, args.lgs_calls, os.environ.get("NP_HOST_THREADS", "4"))},
                              "s_per_call": lgs["s_per_call"], "host_cpu_s_per_mbp": lgs["cpu_s_per_mbp"]}
                if "cpu_baseline" in lgs:
                    out["lgs"]["cpu_baseline"] = lgs["cpu_baseline"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(make_stream, int(args.cpu_sample_mb * 1e6), depth, 424242)
        print(json.dumps(out))
    batch.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
