"""ctypes binding of the REAL reference long-read library (oracle/_ref/nextpolish2.so, built by oracle/Makefile from
the sources under /root/reference).  TEST INFRASTRUCTURE ONLY: the checker for the nextpolish2 path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "..", "oracle", "_ref", "nextpolish2.so")


class ConsensusTrimed(C.Structure):
    _fields_ = [("len", C.c_uint), ("identity", C.c_float), ("seq", C.c_char_p)]


class ConsensusTrimedData(C.Structure):
    _fields_ = [("data", C.POINTER(ConsensusTrimed)), ("i_m", C.c_int)]


class Ref(C.Structure):
    _fields_ = [("n", C.c_char_p), ("s", C.POINTER(C.c_uint32)), ("qv", C.c_void_p), ("qv_l", C.c_uint32),
                ("length", C.c_uint32)]


class Refs(C.Structure):
    _fields_ = [("ref", C.POINTER(Ref)), ("i", C.c_uint32), ("i_m", C.c_uint32)]


def available(path=REF_SO):
    return os.path.exists(path)


def bind(path):
    L = C.CDLL(path)
    L.read_ref.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int]
    L.read_ref.restype = C.POINTER(Refs)
    L.refs_destroy.argtypes = [C.POINTER(Refs)]
    L.ctg_cns_init.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
    L.ctg_cns_init.restype = C.c_void_p
    L.ctg_cns_destroy.argtypes = [C.c_void_p]
    L.ctg_cns_core.argtypes = [C.c_void_p, C.POINTER(Ref), C.c_char_p]
    L.ctg_cns_core.restype = C.POINTER(ConsensusTrimedData)
    L.free_consensus_trimed_data.argtypes = [C.POINTER(ConsensusTrimedData)]
    L.seq2bit1.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_char_p]
    L.bit2seq1.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_char_p]
    return L


def polish(L, fasta, bam_list, window=5000000, read_type=1, split=0, names=None):
    """Runs read_ref + ctg_cns_core over every contig; returns {name: [piece strings]} and the packed words."""
    if names:
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        refs = L.read_ref(fasta.encode(), arr, len(names))
    else:
        refs = L.read_ref(fasta.encode(), None, 0)
    cfg = L.ctg_cns_init(window, read_type, split, 0.8, 0.8, 0.8)
    out = {}
    for i in range(refs.contents.i):
        r = refs.contents.ref[i]
        d = L.ctg_cns_core(cfg, C.byref(r), bam_list.encode())
        pieces = [(C.string_at(d.contents.data[k].seq).decode(), int(d.contents.data[k].len)) for k in range(d.contents.i_m)]
        out[r.n.decode()] = pieces
        L.free_consensus_trimed_data(d)
    L.ctg_cns_destroy(cfg)
    L.refs_destroy(refs)
    return out
