"""The DEVICE code of the pseudo-seed kernel (nextpolish_amd/csrc/np2_poa_dev.h: partial-order alignment of a region's candidates, one wave per
region, two size classes) run on the CPU -- 64 host threads in lockstep with the wave intrinsics supplied by tests/model/np2_poa_emu.cpp --
against the host version (np2_poa.cpp), which the known-answer tests and the fuzz of tests/test_np2_cpu.py pin to the reference's
poa_to_consensus (source/lib/dag.c:658-694).  What this cannot see is what only the compiler and the hardware decide (round 5's endless loop
was a reconvergence matter in the job loop of the kernel); the GPU tests with NP2_POA_CHECK=1 cover that."""
import ctypes as C
import os
import random
import subprocess

import pytest

import np2_strings

HERE = os.path.dirname(os.path.abspath(__file__))
MODEL = os.path.join(HERE, "model")


@pytest.fixture(scope="module")
def libs():
    subprocess.run(["make", "-C", MODEL, "libnp2_model.so", "libnp2_poa_emu.so"], check=True, capture_output=True)
    M = C.CDLL(os.path.join(MODEL, "libnp2_model.so"))
    E = C.CDLL(os.path.join(MODEL, "libnp2_poa_emu.so"))
    E.np2poa_emu.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_int, C.c_char_p, C.c_int]
    return M, E


def emu(E, seqs, cls):
    arr = (C.c_char_p * len(seqs))(*[s.encode("latin1") for s in seqs])
    buf = C.create_string_buffer(100000)
    rc = E.np2poa_emu(arr, len(seqs), cls, buf, 99000)
    return rc, buf.value.decode("latin1")


def region(rng):
    """candidates of one low-quality region the way the long-read path makes them: copies of one string with substitutions, indels and a few
    ambiguity letters"""
    L = rng.choice([6, 18, 30, 45, 60, 90])
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
        seqs.append(("".join(s) or "A")[:126])
    return seqs


@pytest.mark.parametrize("seed", range(6))
def test_device_pseudo_seed_code_equals_the_host_version(libs, seed):
    M, E = libs
    seqs = region(random.Random(9100 + seed))
    want = np2_strings.model_poa(M, seqs)
    done = 0
    for cls in (0, 1):      # 0 = Small (byte indices, table in LDS), 1 = Big (table in scratch, last two rows in LDS)
        rc, got = emu(E, seqs, cls)
        assert rc in (0, 1), (seed, cls, rc)          # 1 = the class gives the region back (graph or table outgrew it)
        if rc == 0:
            assert got == want, (seed, cls, seqs)
            done += 1
    assert done >= 1, "both classes gave the region back"


def test_classes_give_back_what_they_cannot_hold(libs):
    M, E = libs
    rng = random.Random(5)
    long_ = ["".join(rng.choice("ACGT") for _ in range(300)) for _ in range(3)]
    assert emu(E, long_, 1)[0] == 1                                  # strings beyond 255 characters: not even the Big class
    wide = ["".join(rng.choice("ACGT") for _ in range(120)) for _ in range(6)]      # unrelated strings: the graph outgrows 127 nodes
    assert emu(E, wide, 0)[0] == 1
