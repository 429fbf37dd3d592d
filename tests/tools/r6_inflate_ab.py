"""A/B of the device BGZF block decoders on generated BAM bytes (round 6).
usage: r6_inflate_ab.py [qualities: 0 none | 1 uniformly random | 2 binned] [contigs of 2.5 Mb to generate = 80] [modes = lanes,lds,lds96,wave]
Every mode decodes the same blocks in a process of its own (NP1_INFLATE is read once); the output of each must hash to the same value, and
a sample of blocks is checked against zlib."""
import ctypes as C, hashlib, os, struct, subprocess, sys, tempfile, zlib
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", ".."))
from nextpolish_amd import _native as nat


def make_bam(wq, contigs):
    d = tempfile.mkdtemp(prefix="np1inf_")
    st = nat.Stream.synth([2500000] * contigs, depth=30.0, seed=5, with_qual=1 if wq == 1 else 0)
    L0 = nat.lib()
    L0.np1_streams_write_files_q.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    arr = (C.c_void_p * 1)(st.handle)
    L0.np1_streams_write_files_q(arr, 1, os.path.join(d, "g.fa").encode(), os.path.join(d, "r.bam").encode(), 1, 1 if wq == 2 else 0)
    return os.path.join(d, "r.bam")


def run(path):
    buf = open(path, "rb").read()
    p = end = 0
    offs = []
    while p + 18 <= len(buf):
        n = struct.unpack_from("<H", buf, p + 16)[0] + 1
        offs.append((p, n))
        p += n
        end = p
    buf = buf[:end]
    nout = sum(struct.unpack_from("<I", buf, o + n - 4)[0] for o, n in offs)
    L = nat.lib()
    L.np1_debug_inflate_device_prof.argtypes = [C.c_int, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_float)]
    L.np1_debug_inflate_device_prof.restype = C.c_int64
    out = np.zeros(nout + 65536, dtype=np.uint8)
    status = np.zeros(len(offs) + 16, dtype=np.uint32)
    best = 1e30
    for _ in range(4):
        ms = C.c_float(0)
        nb = L.np1_debug_inflate_device_prof(0, buf, len(buf), out.ctypes.data, len(out), status.ctypes.data, len(status), None, C.byref(ms))
        assert nb == len(offs), nat.last_error()
        best = min(best, ms.value)
    rej = int((status[:nb] != 0).sum())
    # a sample of blocks against zlib
    at = 0
    bad = 0
    for k, (o, n) in enumerate(offs):
        isz = struct.unpack_from("<I", buf, o + n - 4)[0]
        if k % 997 == 0 and not status[k]:
            xlen = struct.unpack_from("<H", buf, o + 10)[0]
            want = zlib.decompress(buf[o + 12 + xlen:o + n - 8], -15)
            bad += want != out[at:at + isz].tobytes()
        at += isz
    md5 = hashlib.md5(out[:nout].tobytes()).hexdigest() if not rej else "-"
    print("%-6s dbg=%s %d blocks, %.1f MB -> %.1f MB in %.2f ms = %.1f GB/s out, %.1f GB/s in+out; rejected %d; sample vs zlib: %d differ; md5 %s"
          % (os.environ.get("NP1_INFLATE", "auto"), os.environ.get("NP1_LDS_DBG", "0"), nb, len(buf) / 1e6, nout / 1e6, best, nout / best / 1e6, (nout + len(buf)) / best / 1e6, rej, bad, md5), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        run(sys.argv[2])
        sys.exit(0)
    wq = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    contigs = int(sys.argv[2]) if len(sys.argv) > 2 else 80
    modes = (sys.argv[3] if len(sys.argv) > 3 else "lanes,lds,lds96,wave").split(",")
    path = make_bam(wq, contigs)
    print("qualities mode %d, %d contigs of 2.5 Mb at 30x: %s, %.1f MB" % (wq, contigs, path, os.path.getsize(path) / 1e6), flush=True)
    for m in modes:      # "lds85:3" = NP1_INFLATE=lds85 with NP1_LDS_DBG=3 (timing experiments: parts of the work left out, wrong output)
        mm, _, dbg = m.partition(":")
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], env=dict(os.environ, NP1_INFLATE=mm, NP1_LDS_DBG=dbg or "0"), check=False)
