"""One synthetic contig at a given depth: HIP library vs the compiled reference (identical?) with timings.
usage: np2_depth_check.py <contig_len> <depth> [read_type]"""
import hashlib, json, os, subprocess, sys, tempfile, time
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", "..")); sys.path.insert(0, os.path.join(here, ".."))
from nextpolish_amd import _native as nat
L, depth = int(sys.argv[1]), float(sys.argv[2])
rt = int(sys.argv[3]) if len(sys.argv) > 3 else 1
d = tempfile.mkdtemp(prefix="np2d_")
kw = dict(sub=0.005, ins=0.003, dele=0.003) if rt == 3 else {}
st = nat.Stream.synth_long([L], depth=depth, seed=11, **kw)
fa, bam, fofn = os.path.join(d, "g.fa"), os.path.join(d, "r.bam"), os.path.join(d, "bam.fofn")
st.write_files(fa, bam); st.close()
open(fofn, "w").write(bam + "\n")
res = {}
for name, so in (("hip", os.path.join(here, "..", "..", "nextpolish_amd", "lib", "nextpolish2.so")), ("ref", os.path.join(here, "..", "..", "oracle", "_ref", "nextpolish2.so"))):
    code = ("import sys, json, hashlib, time; sys.path.insert(0, %r); import ref2_binding as rb; L = rb.bind(%r); n = %d\n"
            "if n: rb.polish(L, %r, %r, read_type=%d)\n"
            "t = time.time(); out = rb.polish(L, %r, %r, read_type=%d); dt = time.time() - t\n"
            "print(json.dumps([dt, {k: [[hashlib.md5(p[0].encode()).hexdigest(), p[1]] for p in v] for k, v in out.items()}]))"
            % (os.path.join(here, ".."), so, 1 if name == "hip" else 0, fa, fofn, rt, fa, fofn, rt))
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, NP2_TIMING="1" if name == "hip" else ""))
    if p.returncode != 0:
        print(name, "failed:", p.stderr[-300:]); continue
    res[name] = json.loads(p.stdout.strip().splitlines()[-1])
    if name == "hip":
        print("\n".join(l for l in p.stderr.splitlines() if "np2 window" in l or "np2 host" in l)[-1600:])
    print("%s: %.2f s -> %.2f Mbp/s" % (name, res[name][0], L / res[name][0] / 1e6), flush=True)
if len(res) == 2:
    print("IDENTICAL" if res["hip"][1] == res["ref"][1] else "DIFFERENT")
