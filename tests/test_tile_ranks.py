"""The tiles of one dominant contig over the ranks of a node (nextpolish_amd/nextpolish1.py: write_tile_pieces / join_tile_pieces, DESIGN.md
section 8): every rank polishes tiles rank, rank + world, ... of a contig longer than --tile_bp and leaves each piece as a file, the contig's
joiner concatenates them in tile order.  CPU tests: the file protocol with made-up pieces, and the real thing with the host model in the
place of the device (tests/model: the product's tiling driver and region loader), against the untiled oracle."""
import os
import sys
import threading
import time

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from nextpolish_amd import _native as nat  # noqa: E402
from nextpolish_amd import nextpolish1 as np1  # noqa: E402


def test_pieces_of_all_ranks_join_in_tile_order(tmp_path):
    d = str(tmp_path / "tiles")
    L, T = 10500, 1000                      # 11 tiles, the last one short

    def piece(name, k, n):
        assert n == 11
        return "<%s:%d>" % (name, k)
    for world in (1, 2, 3, 16):
        for rank in reversed(range(world)):      # (any order of arrival)
            np1.write_tile_pieces(piece, d, "ctg/with odd:name", L, T, world, rank)
        got = np1.join_tile_pieces(d, "ctg/with odd:name", L, T, wait_s=1)
        assert got == "".join("<ctg/with odd:name:%d>" % k for k in range(11))
        assert not os.path.exists(np1.tile_piece_dir(d, "ctg/with odd:name"))      # the joiner cleans up


def test_joiner_waits_for_a_late_rank_and_gives_up_after_the_limit(tmp_path):
    d = str(tmp_path / "tiles")
    L, T = 5000, 1000

    def piece(name, k, n):
        return "%d," % k
    np1.write_tile_pieces(piece, d, "c", L, T, 2, 0)

    def late():
        time.sleep(0.5)
        np1.write_tile_pieces(piece, d, "c", L, T, 2, 1)
    t = threading.Thread(target=late)
    t.start()
    assert np1.join_tile_pieces(d, "c", L, T, wait_s=20, poll_s=0.05) == "0,1,2,3,4,"
    t.join()
    np1.write_tile_pieces(piece, d, "c", L, T, 2, 0)             # rank 1 never comes
    with pytest.raises(SystemExit) as e:
        np1.join_tile_pieces(d, "c", L, T, wait_s=0.3, poll_s=0.05)
    assert "tile 1 of c did not arrive" in str(e.value)


def test_shared_contigs_are_the_long_ones_in_block_order():
    lens = {"a": 10, "b": 5000, "c": 100, "d": 7000}
    assert np1.shared_tile_contigs(["d", "a", "b", "c", "zz"], lens, 1000) == ["d", "b"]
    assert np1.tile_count(7000, 1000) == 7 and np1.tile_count(7001, 1000) == 8


def test_two_ranks_with_the_host_model_as_the_device_equal_the_untiled_oracle(tmp_path):
    """write_tile_pieces / join_tile_pieces around tile pieces computed by the product's tiling driver with the host model as the device
    (first_tile = k, stride = number of tiles: what device_tile_piece asks np1_score_chain_tiled for)"""
    import model_binding as mb
    import oracle_binding as ob
    st = nat.Stream.synth([26000, 900], depth=20, seed=4711, read_indel=0.004, softclip_rate=0.05, draft_lower=0.02, weird_rate=0.02)
    fa, bam = str(tmp_path / "g.fa"), str(tmp_path / "r.bam")
    st.write_files(fa, bam)
    want = ob.score_chain(st, 0)
    name, L = st.names[0], int(st.ctg_len[0])
    for tile, halo, world in ((3000, 50, 2), (7001, 5, 3)):
        calls = []

        def piece(n, k, nt):
            calls.append(k)
            return mb.score_chain_tiled_files(fa, bam, n, tile, halo, first_tile=k, tile_stride=nt)[0]
        d = str(tmp_path / ("tiles_%d" % tile))
        for rank in range(world):
            np1.write_tile_pieces(piece, d, name, L, tile, world, rank)
        assert sorted(calls) == list(range(np1.tile_count(L, tile)))      # every tile once, over all ranks
        assert np1.join_tile_pieces(d, name, L, tile, wait_s=1) == want
