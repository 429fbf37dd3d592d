"""Stage timing of the long-read path on a prepared case directory (g.fa + bam.fofn): NP2_TIMING lines of the second
call + wall time, for 1..N host threads.  usage: np2_stage_time.py <case dir> [read_type]"""
import os, sys, time, subprocess
here = os.path.dirname(os.path.abspath(__file__))
case = sys.argv[1]
rt = int(sys.argv[2]) if len(sys.argv) > 2 else 1
code = """
import sys, time, os
sys.path.insert(0, %r)
import ref2_binding as rb
G = rb.bind(%r)
fa, fofn = %r, %r
rb.polish(G, fa, fofn, read_type=%d)
sys.stderr.write("==== second call\\n")
t = time.time(); out = rb.polish(G, fa, fofn, read_type=%d); dt = time.time() - t
n = sum(len(p[0]) for v in out.values() for p in v)
sys.stderr.write("==== wall %%.3f s, %%d bp -> %%.2f Mbp/s\\n" %% (dt, n, n / dt / 1e6))
""" % (os.path.join(here, "..", "tests"), os.path.join(here, "..", "nextpolish_amd", "lib", "nextpolish2.so"),
       os.path.join(case, "g.fa"), os.path.join(case, "bam.fofn"), rt, rt)
for th in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["1", "8"]):
    env = dict(os.environ, NP2_TIMING="1", NP_HOST_THREADS=th, NP_IO_THREADS=th)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    err = p.stderr
    print("#### host threads %s (rc %d)" % (th, p.returncode))
    print(err[err.find("==== second call"):] if "==== second call" in err else err[-3000:])
