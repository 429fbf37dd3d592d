"""Phase clocks of the device BGZF decoder on generated BAM bytes.  usage: np1_inflate_prof.py [with_qual=0] [MB of BAM to take=64]"""
import ctypes as C, os, sys, tempfile, zlib, struct
import numpy as np
here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", ".."))
from nextpolish_amd import _native as nat
WQ = int(sys.argv[1]) if len(sys.argv) > 1 else 0
TAKE = int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else 64000000
src = sys.argv[3] if len(sys.argv) > 3 else None
if src is None:
    d = tempfile.mkdtemp(prefix="np1inf_")
    st = nat.Stream.synth([2500000] * int(os.environ.get("NP1_PROF_CONTIGS", "4")), depth=30.0, seed=5, with_qual=1 if WQ == 1 else 0)
    L0 = nat.lib()
    L0.np1_streams_write_files_q.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.c_int]
    arr = (C.c_void_p * 1)(st.handle)
    L0.np1_streams_write_files_q(arr, 1, os.path.join(d, "g.fa").encode(), os.path.join(d, "r.bam").encode(), 1, 1 if WQ == 2 else 0)   # WQ 2: binned qualities
    src = os.path.join(d, "r.bam")
buf = open(src, "rb").read()
# whole blocks only
p, end = 0, 0
while p + 18 <= len(buf) and p < TAKE:
    n = struct.unpack_from("<H", buf, p + 16)[0] + 1
    p += n
    end = p
buf = buf[:end]
want = sum(struct.unpack_from("<I", buf, o + n - 4)[0] for o, n in [(0, 0)][:0]) if False else None
L = nat.lib()
L.np1_debug_inflate_device_prof.argtypes = [C.c_int, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_float)]
L.np1_debug_inflate_device_prof.restype = C.c_int64
out = np.zeros(len(buf) * 12 + 65536, dtype=np.uint8)
status = np.zeros(len(buf) // 26 + 16, dtype=np.uint32)
for mode in ("plain", "plain", "prof"):
    prof = np.zeros(8, dtype=np.uint64)
    ms = C.c_float(0)
    nb = L.np1_debug_inflate_device_prof(0, buf, len(buf), out.ctypes.data, len(out), status.ctypes.data, len(status), prof.ctypes.data if mode == "prof" else None, C.byref(ms))
    assert nb > 0, nat.last_error()
    nout = 0
    o = 0
    for _ in range(nb):
        n = struct.unpack_from("<H", buf, o + 16)[0] + 1
        nout += struct.unpack_from("<I", buf, o + n - 4)[0]
        o += n
    print("%s: %d blocks, %.1f MB -> %.1f MB in %.2f ms = %.1f GB/s out; rejected %d" % (mode, nb, len(buf) / 1e6, nout / 1e6, ms.value, nout / ms.value / 1e6, int((status[:nb] != 0).sum())))
    if mode == "prof":
        t, dcd, fl, tok, grp, rnd, mt, mb = [int(x) for x in prof]
        print("per block: tables %.0f k cycles, decode %.0f k, flush %.0f k | tokens %.0f, groups %.0f, rounds/group %.2f, matches %.0f (%.1f bytes each), cycles/token decode %.0f flush/group %.0f"
              % (t / nb / 1e3, dcd / nb / 1e3, fl / nb / 1e3, tok / nb, grp / nb, rnd / max(1, grp), mt / nb, mb / max(1, mt), dcd / max(1, tok), fl / max(1, grp)))
