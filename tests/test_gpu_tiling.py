"""Intra-contig tiling in the product (nextpolish_amd/csrc/np1_tile.cpp): a contig polished as independent tiles, each
reading its own region of the BAM through the index, joins to exactly what the untiled pass gives -- and to the CPU oracle.  The reference
polishes a contig of any length up to 2^31 in one score_chain call (source/lib/scorechain.c:3-15, source/nextPolish:101-102); the scheme's
proof by fuzz is the host model's (tests/test_model.py::test_model_intra_contig_tiling_*)."""
import ctypes as C
import os
import subprocess

import pytest

from nextpolish_amd import _native as nat
from nextpolish_amd.device import Context
import oracle_binding as ob
from conftest import ROOT, parse_cli_fasta

pytestmark = pytest.mark.gpu


def tiled(ctx, fa, bam, name, tile_bp, halo_bp, first=0, stride=1):
    L = nat.lib()
    L.np1_score_chain_tiled.restype = C.c_int
    L.np1_score_chain_tiled.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(nat.Configure), C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
    L.np1_free_string.argtypes = [C.c_void_p]
    cfg = nat.default_config()
    out, n, st = C.c_void_p(), C.c_int64(0), (C.c_uint64 * 4)()
    rc = L.np1_score_chain_tiled(ctx.handle, fa.encode(), bam.encode(), name.encode(), C.byref(cfg), tile_bp, halo_bp, first, stride, C.byref(out), C.byref(n), st)
    assert rc == 0, nat.last_error()
    s = C.string_at(out, n.value).decode()
    L.np1_free_string(out)
    return s, dict(tiles=st[0], recomputed=st[1], records=st[2], largest=st[3])


def test_tiles_of_every_size_join_to_the_oracle(tmp_path):
    """tiles from a few hundred bases to a fraction of the contig, halos from one base up (a halo without a single-state slot makes the
    tile run again with twice the halo), indel-rich reads, soft clips, lower case: always the oracle's string"""
    st = nat.Stream.synth([60000, 9000], depth=25, seed=4242, read_indel=0.004, softclip_rate=0.05, draft_lower=0.02, weird_rate=0.01)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "r.bam")
    st.write_files(fa, bam)
    want = [ob.score_chain(st, i) for i in range(st.n_contigs)]
    ctx = Context(0)
    redo = 0
    for tile, halo in ((700, 1), (5000, 40), (20000, 300), (59999, 150), (100000, 50)):
        for i, n in enumerate(st.names):
            got, info = tiled(ctx, fa, bam, n, tile, halo)
            assert got == want[i], (tile, halo, n, info)
            assert info["tiles"] == -(-int(st.ctg_len[i]) // tile)
            redo += info["recomputed"]
    assert redo > 0          # the one-base halos really were too small somewhere
    # tiles dealt over two ranks: the pieces of the ranks, concatenated tile by tile, are the contig
    tile = 8000
    a, ia = tiled(ctx, fa, bam, st.names[0], tile, 200, 0, 2)
    b, ib = tiled(ctx, fa, bam, st.names[0], tile, 200, 1, 2)
    assert ia["tiles"] + ib["tiles"] == -(-int(st.ctg_len[0]) // tile) and len(a) + len(b) == len(want[0])
    ctx.close()


def test_cli_with_tiling_equals_cli_without(tmp_path):
    """`nextpolish1 scorechain` with NP1_TILE_BP: a 24 Mb contig in 8 tiles of 3 Mb between short contigs that take the batched pipe, output
    identical to the untiled run, byte for byte, in index order"""
    st = nat.Stream.synth([300000, 24000000, 150000, 2000000], depth=30, seed=4243)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "r.bam")
    st.write_files(fa, bam)
    exe = os.path.join(ROOT, "nextpolish_amd", "bin", "nextpolish1")
    plain = subprocess.run([exe, "scorechain", fa, bam], capture_output=True, text=True)
    assert plain.returncode == 0, plain.stderr[-800:]
    t = subprocess.run([exe, "scorechain", fa, bam], capture_output=True, text=True, env=dict(os.environ, NP1_TILE_BP="3000000", NP1_TIMING="1"))
    assert t.returncode == 0, t.stderr[-800:]
    import re
    m = re.findall(r"\[np1 tiles\] (\S+): (\d+) bases in (\d+) tiles", t.stderr)
    assert len(m) == 1 and m[0][0] == st.names[1] and int(m[0][2]) == -(-int(m[0][1]) // 3000000) >= 8, t.stderr[-800:]
    assert t.stdout == plain.stdout
    assert list(parse_cli_fasta(t.stdout)) == list(st.names)


def test_python_caller_with_tile_bp_equals_the_untiled_caller(tmp_path):
    """nextpolish1.py -t 1 --tile_bp: the long contig in tiles, the short ones through the batched pipe, same FASTA as without the option"""
    import sys
    st = nat.Stream.synth([200000, 6000000, 90000], depth=30, seed=4244)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "r.bam")
    st.write_files(fa, bam)
    outs = []
    for extra in ([], ["--tile_bp", "900k", "--tile_halo", "500"]):
        out = str(tmp_path / ("o%d.fa" % len(outs)))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "nextpolish_amd", "nextpolish1.py"), "-g", fa, "-t", "1", "-s", bam, "-o", out] + extra,
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-800:]
        outs.append(open(out).read())
    assert outs[0] == outs[1] and outs[0].count(">") == 3


def test_two_ranks_share_the_tiles_of_the_long_contigs(tmp_path):
    """nextpolish1.py --world 2 --tile_bp: two caller processes on the one GPU; contigs longer than --tile_bp are shared tile by tile (rank r
    takes tiles r, r + 2, ... in ONE np1_tiler_run per contig, the pieces meet in --tile_dir under the launch's token), the rest is dealt whole;
    the two -o parts together are the untiled single-rank run, and a rank that dies ends its peer's wait at once (VERDICT r4 missing 5)."""
    import sys
    st = nat.Stream.synth([200000, 6000000, 90000, 2500000, 30000], depth=30, seed=8128)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "r.bam")
    st.write_files(fa, bam)
    caller = os.path.join(ROOT, "nextpolish_amd", "nextpolish1.py")

    def _records(path):
        out, name = {}, None
        for line in open(path):
            if line.startswith(">"):
                name = line[1:].split()[0]
                out[name] = ""
            else:
                out[name] += line.strip()
        return out
    one = str(tmp_path / "one.fa")
    subprocess.run([sys.executable, caller, "-g", fa, "-t", "1", "-s", bam, "-o", one], check=True)
    tiles = str(tmp_path / "tiles")
    ps = [subprocess.Popen([sys.executable, caller, "-g", fa, "-t", "1", "-s", bam, "-o", str(tmp_path / ("part%d.fa" % r)), "--world", "2", "--rank", str(r),
                            "--device", "0", "--tile_bp", "1000000", "--tile_dir", tiles, "--tile_wait", "600"]) for r in range(2)]
    assert [p.wait() for p in ps] == [0, 0]
    want, got = _records(one), {}
    for r in range(2):
        part = _records(str(tmp_path / ("part%d.fa" % r)))
        assert not set(part) & set(got), "a contig in both parts"
        got.update(part)
    assert got == want, [n for n in want if got.get(n) != want[n]]
    assert [f for f in os.listdir(tiles) if not f.startswith("FAILED")] == [], "pieces left behind"
    # rank 1 cannot run (bad device): rank 0 must not wait --tile_wait seconds for its pieces
    import time
    t0 = time.time()
    p0 = subprocess.Popen([sys.executable, caller, "-g", fa, "-t", "1", "-s", bam, "-o", str(tmp_path / "p0.fa"), "--world", "2", "--rank", "0", "--device", "0",
                           "--tile_bp", "1000000", "--tile_dir", tiles, "--tile_wait", "600"], stderr=subprocess.PIPE, text=True)
    p1 = subprocess.run([sys.executable, caller, "-g", fa, "-t", "1", "-s", bam, "-o", str(tmp_path / "p1.fa"), "--world", "2", "--rank", "1", "--device", "99",
                         "--tile_bp", "1000000", "--tile_dir", tiles, "--tile_wait", "600"], capture_output=True, text=True)
    assert p1.returncode != 0
    err0 = p0.communicate(timeout=300)[1]
    assert p0.returncode != 0 and "will not arrive" in err0 and time.time() - t0 < 200, err0[-600:]
