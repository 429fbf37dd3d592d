#!/usr/bin/env python
"""A kmer_count round (task 2) at the scale of BASELINE config 5: ONE chromosome-sized contig -- 250 Mb at 30x, 50 M records with
Illumina-like binned qualities, a sorted BAM of its own -- FROM FILES through `nextpolish1 kmercount` (device-side BGZF inflate +
record split, the reference's region iterator replayed on the index and the records' virtual offsets the ingest brings down), against
the CPU oracle run on the same files in replay mode (tests/oracle_binding.py:from_files).  0.5 % of the draft is lower case, i.e.
tens of thousands of k-mer regions.  Prints one JSON line; exit code 1 on a mismatch.
usage: check_config5_kmer.py [mb] [--quick]"""
import ctypes as C
import hashlib
import json
import os
import shutil
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
import oracle_binding as ob  # noqa: E402
from conftest import parse_cli_fasta  # noqa: E402


def md5(s):
    return hashlib.md5(s.encode() if isinstance(s, str) else s).hexdigest()


def main():
    quick = "--quick" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    mb = float(args[0]) if args else (12.0 if quick else 250.0)
    info = {"mismatches": 0}
    t0 = time.time()
    st = nat.Stream.synth([int(mb * 1e6)], depth=30.0, seed=79, with_qual=1, draft_lower=0.005, prefix="k")
    info.update({"draft_bp": int(st.ctg_len[0]), "records": st.n_reads, "synth_seconds": round(time.time() - t0, 1)})
    d = tempfile.mkdtemp(prefix="np1c5k_")
    fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
    L = nat.lib()
    L.np1_streams_write_files_q.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    arr = (C.c_void_p * 1)(st.handle)
    t0 = time.time()
    if L.np1_streams_write_files_q(arr, 1, fa.encode(), bam.encode(), 1, 1) != 0:
        raise SystemExit(nat.last_error())
    info["write_seconds"] = round(time.time() - t0, 1)
    st.close()
    cfgp = L.config_init(fa.encode(), bam.encode(), None)
    ocfg = ob.default_config(read_tlen=cfgp.contents.read_tlen, read_len=cfgp.contents.read_len)
    L.config_destory(cfgp)
    ex = ThreadPoolExecutor(1)
    t_or = time.time()
    f_or = ex.submit(lambda: ob.from_files("kmer_count", fa, bam, ocfg))       # host loader + oracle with the iterator replayed, beside the GPU work
    exe = os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1")
    t0 = time.time()
    p = subprocess.run([exe, "kmercount", fa, bam], capture_output=True, text=True, env=dict(os.environ, NP1_TIMING="1"))
    info["cli_seconds"] = round(time.time() - t0, 2)
    if p.returncode != 0:
        raise SystemExit("nextpolish1 kmercount failed: " + p.stderr[-500:])
    cli = parse_cli_fasta(p.stdout)
    want = f_or.result()
    info["oracle_seconds"] = round(time.time() - t_or, 1)
    info.update({"bam_mb": round(os.path.getsize(bam) / 1e6, 1), "contigs": len(cli)})
    for n, s in want.items():
        if s is None or n not in cli or md5(cli[n]) != md5(s):
            info["mismatches"] += 1
        else:
            info["polished_bp"] = len(s)
            info["lower_case_out"] = sum(1 for c in s if c.islower())
    shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(info))
    return 1 if info["mismatches"] else 0


if __name__ == "__main__":
    sys.exit(main())
