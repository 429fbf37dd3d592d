"""Where the time of `nextpolish1 snpphase` from files goes: stage clocks (NP1_TIMING) of the CLI on the workload of bench.py's snp_phase leg.
usage: np1_phase_e2e_prof.py [Mb=20] [qual_model=0|1] [batch_mb ...]"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nextpolish_amd import _native as nat   # noqa: E402


def main():
    mb = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    qm = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    batches = [float(x) for x in sys.argv[3:]] or [mb + 1, 4.0]
    n_ctg = max(1, int(mb // 4))
    sr, lr = nat.Stream.synth_diploid([int(mb * 1e6 / n_ctg)] * n_ctg, seed=9090, sr_holes=2)
    bp = int(sr.ctg_len.sum())
    td = tempfile.mkdtemp(prefix="np1phase_prof_")
    fa, s_bam, l_bam = os.path.join(td, "g.fa"), os.path.join(td, "sr.bam"), os.path.join(td, "lr.bam")
    if qm:
        import ctypes as C
        L = nat.lib()
        L.np1_streams_write_files_q.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
        arr = (C.c_void_p * 1)(sr.handle)
        assert L.np1_streams_write_files_q(arr, 1, fa.encode(), s_bam.encode(), 1, qm) == 0
    else:
        sr.write_files(fa, s_bam, 1)
    lr.write_files(os.path.join(td, "l.fa"), l_bam, 1)
    print("files: short %.1f MB, long %.1f MB, %d + %d records" % (os.path.getsize(s_bam) / 1e6, os.path.getsize(l_bam) / 1e6, sr.n_reads, lr.n_reads), flush=True)
    exe = os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1")
    for bm in batches:
        for rep in range(2):
            env = dict(os.environ, NP1_TIMING="1", NP1_BATCH_BP=str(int(bm * 1e6)))
            t0 = time.time()
            q = subprocess.run([exe, "snpphase", fa, s_bam, l_bam], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
            dt = time.time() - t0
            print("batch %.1f Mb run %d: %.2f s = %.1f Mbp/s (rc %d)" % (bm, rep, dt, bp / 1e6 / dt, q.returncode), flush=True)
            if rep == 1:
                print(q.stderr.decode()[-3000:], flush=True)


if __name__ == "__main__":
    main()
