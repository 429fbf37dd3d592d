"""Inputs and reference bindings for the two string algorithms of the long-read path (test helper): partial-order
consensus (reference: source/lib/dag.c poa_to_consensus) and banded O(ND) alignment (source/lib/align.c align)."""
import ctypes as C


class Seq(C.Structure):   # struct seq_ (ctg_cns.h:83-89)
    _fields_ = [("order", C.c_uint16), ("kscore", C.c_uint16), ("len", C.c_uint32), ("seq", C.c_char_p)]


class Aln(C.Structure):   # alignment (ctg_cns.h:126-138)
    _fields_ = [("shift", C.c_uint), ("aln_len", C.c_uint), ("max_aln_len", C.c_uint), ("aln_t_s", C.c_uint), ("aln_t_e", C.c_uint),
                ("aln_t_len", C.c_uint), ("aln_q_s", C.c_uint), ("aln_q_e", C.c_uint), ("aln_q_len", C.c_uint),
                ("q_aln_str", C.c_char_p), ("t_aln_str", C.c_char_p)]


def mutate(rng, s, rate, biggap=False):
    out, i = [], 0
    while i < len(s):
        if biggap and rng.random() < 0.002:
            if rng.random() < 0.5:
                i += rng.randint(100, 400)
                continue
            out += [rng.choice("ACGT") for _ in range(rng.randint(100, 400))]
        x = rng.random()
        if x < rate / 3:
            i += 1
            continue
        if x < 2 * rate / 3:
            out.append(rng.choice("ACGT"))
        if x < rate:
            out.append(rng.choice("ACGT"))
            i += 1
            continue
        out.append(s[i])
        i += 1
    return "".join(out) or "A"


def poa_case(rng):
    L = rng.choice([5, 30, 120, 400])
    base = "".join(rng.choice("ACGT") for _ in range(L))
    n = rng.randint(2, 6)
    seqs = [mutate(rng, base, rng.choice([0.02, 0.1, 0.25])) for _ in range(n)]
    if rng.random() < 0.2:
        seqs[rng.randrange(n)] = seqs[0] + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 40)))
    if rng.random() < 0.2:
        seqs[rng.randrange(n)] = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 30))) + seqs[0]
    return seqs


def align_case(rng, seed):
    L = rng.choice([4, 20, 150, 600, 1500])
    t = "".join(rng.choice("ACGT") for _ in range(L))
    q = mutate(rng, t, rng.choice([0.0, 0.02, 0.1, 0.3]), biggap=seed % 5 == 0)
    if seed % 7 == 0:
        q = q[:len(q) // 2]
    return q, t


def ref_poa(R, seqs):
    R.poa_to_consensus.argtypes = [C.POINTER(Seq), C.c_int]
    R.poa_to_consensus.restype = C.c_void_p
    arr = (Seq * len(seqs))()
    keep = [s.encode() for s in seqs]
    for i, s in enumerate(keep):
        arr[i].len = len(s)
        arr[i].seq = s
    return C.string_at(R.poa_to_consensus(arr, len(seqs))).decode("latin1")


def ref_align(R, q, t):
    R.align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(Aln), C.POINTER(C.c_int), C.c_void_p]
    R.malloc_vd.argtypes = [C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_void_p), C.c_uint64]
    ql, tl = len(q), len(t)
    mem = ql + tl + 2
    max_mem_d = int(mem * 0.4) + 1
    V, D = C.POINTER(C.c_int)(), C.c_void_p()
    R.malloc_vd(C.byref(V), C.byref(D), max_mem_d)
    C.memset(V, 0, max_mem_d * 2 * 4)
    tb, qb = C.create_string_buffer(mem + 10), C.create_string_buffer(mem + 10)
    a = Aln()
    a.aln_len, a.aln_t_s, a.aln_t_e, a.shift = 0, 0, tl, 0
    a.t_aln_str, a.q_aln_str = C.cast(tb, C.c_char_p), C.cast(qb, C.c_char_p)
    R.align(q.encode(), ql, t.encode(), tl, C.byref(a), V, D)
    n = a.aln_len
    return n, tb.raw[:n].decode("latin1"), qb.raw[:n].decode("latin1"), a.aln_t_len, a.aln_q_len


def model_poa(M, seqs):
    M.np2m_poa.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_int]
    arr = (C.c_char_p * len(seqs))(*[s.encode() for s in seqs])
    buf = C.create_string_buffer(100000)
    M.np2m_poa(arr, len(seqs), buf, 100000)
    return buf.value.decode("latin1")


def model_align(M, q, t):
    M.np2m_align.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_int)]
    cap = len(q) + len(t) + 16
    ot, oq, lens = C.create_string_buffer(cap), C.create_string_buffer(cap), (C.c_int * 2)()
    n = M.np2m_align(q.encode(), len(q), t.encode(), len(t), ot, oq, cap, lens)
    return n, ot.raw[:n].decode("latin1"), oq.raw[:n].decode("latin1"), lens[0], lens[1]
