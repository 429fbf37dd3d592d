import os, subprocess, sys, tempfile, time, ctypes as C
sys.path.insert(0, os.getcwd())
from nextpolish_amd import _native as nat
from concurrent.futures import ThreadPoolExecutor
d = tempfile.mkdtemp(prefix="np1tl_")
with ThreadPoolExecutor(8) as ex:
    sts = list(ex.map(lambda k: nat.Stream.synth([2500000] * 10, depth=30.0, seed=100 + k, with_qual=0, prefix="b%dc" % k), range(16)))
fa, bam = os.path.join(d, "g.fa"), os.path.join(d, "r.bam")
L = nat.lib()
L.np1_streams_write_files_q.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
arr = (C.c_void_p * len(sts))(*[s.handle for s in sts])
assert L.np1_streams_write_files_q(arr, len(sts), fa.encode(), bam.encode(), 1, 1) == 0
exe = os.path.join("nextpolish_amd", "bin", "nextpolish1")
for rep in range(2):
    t0 = time.time()
    p = subprocess.Popen([exe, "scorechain", fa, bam], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, NP1_TIMING="1"))
    ev = []
    for ln in p.stderr:
        ev.append((time.time() - t0, ln.decode().rstrip()[:110]))
    p.wait()
    t_end = time.time() - t0
    print("==== run %d: %.3f s" % (rep, t_end))
    keys = [e for e in ev if "lanes open" in e[1] or "np1 pipe] batch" in e[1] and "lane" in e[1] or "total time" in e[1]]
    print("  first lines:"); [print("   %.3f %s" % e) for e in ev[:4]]
    print("  first batch done:"); [print("   %.3f %s" % e) for e in keys[1:3]]
    print("  last:"); [print("   %.3f %s" % e) for e in ev[-3:]]
    print("  exit at %.3f" % t_end)
