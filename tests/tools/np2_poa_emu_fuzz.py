"""The device code of the pseudo-seed kernel (np2_poa_dev.h, both size classes) in the 64-thread lockstep emulator (tests/model/np2_poa_emu.cpp)
against the host version on random regions -- the long form of tests/test_np2_poa_emu.py.  CPU only.
usage: np2_poa_emu_fuzz.py [first_seed=0] [n=200]"""
import ctypes as C
import os
import random
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
T = os.path.join(here, "..")
sys.path.insert(0, T)
import np2_strings  # noqa: E402

MODEL = os.path.join(T, "model")
subprocess.run(["make", "-C", MODEL, "libnp2_model.so", "libnp2_poa_emu.so"], check=True, capture_output=True)
M = C.CDLL(os.path.join(MODEL, "libnp2_model.so"))
E = C.CDLL(os.path.join(MODEL, "libnp2_poa_emu.so"))
E.np2poa_emu.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_char_p, C.c_int]


def emu(seqs, cls):
    arr = (C.c_char_p * len(seqs))(*[s.encode("latin1") for s in seqs])
    buf = C.create_string_buffer(100000)
    return E.np2poa_emu(arr, len(seqs), cls, buf, 99000), buf.value.decode("latin1")


first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad, done, back = 0, [0, 0], [0, 0]
for seed in range(first, first + n):
    rng = random.Random(seed)
    if seed % 3 == 0:
        seqs = np2_strings.poa_case(rng)
    else:
        L = rng.randint(5, 125)
        base = "".join(rng.choice("ACGT") for _ in range(L))
        seqs = []
        for _ in range(rng.randint(2, 7)):
            s = []
            for ch in base:
                r = rng.random()
                if r < 0.04:
                    continue
                if r < 0.08:
                    s.append(rng.choice("ACGT"))
                if r < 0.12:
                    s.append(rng.choice("ACGTMRWN"))
                    continue
                s.append(ch)
            seqs.append("".join(s) or "A")
    want = np2_strings.model_poa(M, seqs)
    for cls in (0, 1):
        if cls == 0 and (len(seqs) > 8 or max(map(len, seqs)) > 126):
            continue
        rc, got = emu(seqs, cls)
        if rc == 1:
            back[cls] += 1
        elif rc != 0 or got != want:
            bad += 1
            print("DIFFERENT seed", seed, "class", cls, "rc", rc, flush=True)
        else:
            done[cls] += 1
print("%d regions: Small equal %d, gave back %d; Big equal %d, gave back %d; different %d" % (n, done[0], back[0], done[1], back[1], bad))
