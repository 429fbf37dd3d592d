#!/usr/bin/env python
"""BASELINE config 5 scale on one MI355X (the 3 Gb metric shape, exercised at the sizes its pieces have):
  (1) ONE chromosome-sized contig -- 250 Mb at 30x, 50 M records, one HBM batch -- through score_chain (resident batch), the polished
      string against the CPU oracle;
  (2) a multi-batch slice FROM FILES: contigs of 60 / 30 / 20 / 3 Mb at 30x written as one sorted BAM with Illumina-like binned
      qualities, polished by the CLI (`nextpolish1 scorechain`, device-side BGZF inflate + record split, NP1_BATCH_BP = 16 Mb so
      every large contig is a batch of its own and the stream of batches is what the 3 Gb run sees), every contig against the oracle.
The oracle runs of both parts share the host cores (one thread per contig).  Prints one JSON line; exit code 1 on a mismatch.
usage: check_config5.py [big_mb] [--quick]"""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault("NP_HOST_THREADS", str(len(os.sched_getaffinity(0))))
from nextpolish_amd import _native as nat  # noqa: E402
from nextpolish_amd.device import Context  # noqa: E402
import oracle_binding as ob  # noqa: E402
from conftest import parse_cli_fasta  # noqa: E402


def md5(s):
    return hashlib.md5(s.encode() if isinstance(s, str) else s).hexdigest()


def main():
    quick = "--quick" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    big_mb = float(args[0]) if args else (20.0 if quick else 250.0)
    slice_lens = [6000000, 3000000, 2000000, 300000] if quick else [60000000, 30000000, 20000000, 3000000]
    info = {"mismatches": 0}
    t0 = time.time()
    big = nat.Stream.synth([int(big_mb * 1e6)], depth=30.0, seed=77)
    sl = nat.Stream.synth(slice_lens, depth=30.0, seed=78, prefix="s")
    info["synth_seconds"] = round(time.time() - t0, 1)
    info["big"] = {"draft_bp": int(big.ctg_len[0]), "records": big.n_reads}
    ex = ThreadPoolExecutor(1 + len(slice_lens))
    t_or = time.time()
    f_big = ex.submit(lambda: md5(ob.score_chain(big, 0)))                 # (ctypes releases the GIL: the oracle runs beside the GPU work)
    f_sl = [ex.submit(lambda i=i: md5(ob.score_chain(sl, i))) for i in range(sl.n_contigs)]
    # ---- (1) the chromosome-sized contig, one resident batch
    ctx = Context(0)
    b = ctx.upload(big)
    cfg = nat.default_config()
    b.score_chain(cfg)
    ms = b.score_chain(cfg, timed=True)
    got_big = b.results()[0]
    info["big"].update({"stage_ms": {k: round(v, 2) for k, v in ms.items()}, "polished_bp": len(got_big), "md5": md5(got_big)})
    b.close()
    ctx.close()
    # ---- (2) the slice from files through the CLI
    d = tempfile.mkdtemp(prefix="np1c5_")
    fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
    L = nat.lib()
    L.np1_streams_write_files_q.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    arr = (C.c_void_p * 1)(sl.handle)
    t0 = time.time()
    if L.np1_streams_write_files_q(arr, 1, fa.encode(), bam.encode(), 1, 1) != 0:
        raise SystemExit(nat.last_error())
    t_write = time.time() - t0
    exe = os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1")
    t0 = time.time()
    p = subprocess.run([exe, "scorechain", fa, bam], capture_output=True, text=True, env=dict(os.environ, NP1_BATCH_BP="16000000", NP1_TIMING="1"))
    t_cli = time.time() - t0
    if p.returncode != 0:
        raise SystemExit("nextpolish1 scorechain failed: " + p.stderr[-500:])
    cli = parse_cli_fasta(p.stdout)
    n_batches = sum(1 for ln in p.stderr.splitlines() if "staged on the host" in ln)
    info["slice"] = {"draft_bp": int(sum(int(x) for x in sl.ctg_len)), "records": sl.n_reads, "contigs": sl.n_contigs, "batches": n_batches,
                     "bam_mb": round(os.path.getsize(bam) / 1e6, 1), "bam_bytes_per_record": round(os.path.getsize(bam) / sl.n_reads, 1),
                     "write_seconds": round(t_write, 1), "cli_seconds": round(t_cli, 2)}
    # ---- the oracle's verdicts
    want_big = f_big.result()
    want_sl = [f.result() for f in f_sl]
    info["oracle_seconds"] = round(time.time() - t_or, 1)
    if md5(got_big) != want_big:
        info["mismatches"] += 1
    for i, n in enumerate(sl.names):
        if n not in cli or md5(cli[n]) != want_sl[i]:
            info["mismatches"] += 1
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(info))
    return 1 if info["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
