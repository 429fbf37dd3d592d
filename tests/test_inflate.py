"""The BGZF block decoder (nextpolish_amd/csrc/np_inflate.cpp, own raw-DEFLATE implementation) against zlib: every block
type (stored, fixed, dynamic), every compression level and strategy, data from constant to incompressible, sizes from 0 to
the 64 KiB of a BGZF block and beyond, and rejection (never a wrong answer) of damaged streams."""
import ctypes as C
import random
import zlib

import numpy as np
import pytest

from nextpolish_amd import _native as nat


DECODER = "np1_debug_inflate"      # the host decoder; the tests below run a second time on np1_debug_inflate_lane


def _inflate(raw, n):
    L = nat.lib()
    fn = getattr(L, DECODER)
    fn.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
    fn.restype = C.c_int
    out = C.create_string_buffer(max(1, n) + 16)
    ok = fn(raw, len(raw), out, n)
    return bool(ok), out.raw[:n]


def _deflate(data, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    return c.compress(data) + c.flush()


def _samples():
    rng = random.Random(5)
    nrng = np.random.default_rng(5)
    yield b""
    yield b"A"
    yield b"ACGT" * 16384
    yield bytes(65536)
    yield bytes(rng.randrange(256) for _ in range(70000))                      # incompressible: stored blocks at level 0, near-stored above
    yield bytes(rng.choice(b"ACGT") for _ in range(65280))                     # sequence-like
    yield nrng.integers(0, 4, 65280, dtype=np.uint8).tobytes()
    yield (b"".join(bytes([rng.randrange(256)]) * rng.randrange(1, 400) for _ in range(600)))[:65536]   # long runs: distance-1 copies
    yield bytes((i * i) & 0xff for i in range(65536))
    yield b"".join(rng.choice([b"the ", b"quick ", b"brown ", b"fox ", b"jumps "]) for _ in range(12000))
    yield nrng.integers(0, 256, 300, dtype=np.uint8).tobytes() * 200          # far matches (distance up to 300 * n)
    yield nrng.normal(128, 3, 65536).astype(np.uint8).tobytes()               # skewed alphabet: long Huffman codes
    for n in (1, 2, 3, 7, 8, 9, 257, 258, 259, 32767, 32768, 32769, 65535):
        yield bytes(rng.choice(b"ACGTN") for _ in range(n))


@pytest.mark.parametrize("level", [0, 1, 4, 6, 9])
def test_matches_zlib_on_every_block_kind(level):
    for data in _samples():
        for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
            raw = _deflate(data, level, strategy)
            ok, got = _inflate(raw, len(data))
            assert ok, "level %d strategy %d len %d rejected" % (level, strategy, len(data))
            assert got == data


def test_multi_block_streams_and_flushes():
    rng = random.Random(9)
    for _ in range(20):
        c = zlib.compressobj(rng.choice([1, 6, 9]), zlib.DEFLATED, -15)
        parts, raw = [], b""
        for _ in range(rng.randrange(1, 8)):
            p = bytes(rng.choice(b"ACGT") for _ in range(rng.randrange(0, 9000)))
            parts.append(p)
            raw += c.compress(p) + c.flush(rng.choice([zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_NO_FLUSH]))   # sync flushes add empty stored blocks
        raw += c.flush()
        data = b"".join(parts)
        ok, got = _inflate(raw, len(data))
        assert ok and got == data


def test_damaged_or_mismatched_streams_are_never_wrong():
    rng = random.Random(3)
    data = bytes(rng.choice(b"ACGT") for _ in range(40000))
    raw = _deflate(data, 6)
    assert _inflate(raw, len(data) - 1)[0] is False       # declared size too small
    assert _inflate(raw, len(data) + 1)[0] is False       # ... too large
    assert _inflate(raw[:len(raw) // 2], len(data))[0] is False   # truncated
    for _ in range(300):                                    # random corruption: rejected, or (rarely) still the right bytes
        b = bytearray(raw)
        for _ in range(rng.randrange(1, 4)):
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        ok, got = _inflate(bytes(b), len(data))
        if ok:
            try:
                want = zlib.decompress(bytes(b), -15)
            except zlib.error:
                want = None
            assert want is not None and got == want[:len(data)] and len(want) == len(data)


def test_block_crc32_equals_zlib():
    """The reader checks the gzip CRC of every block like htslib does; the CRC comes from carry-less-multiply folding
    (nextpolish_amd/csrc/np_crc32.h) where the CPU has PCLMULQDQ, zlib's table code otherwise and for the tail bytes."""
    import ctypes as C
    import os
    import random
    from nextpolish_amd import _native as nat
    L = nat.lib()
    L.np1_debug_crc32.argtypes = [C.c_char_p, C.c_uint64]
    L.np1_debug_crc32.restype = C.c_uint32
    rng = random.Random(7)
    blob = os.urandom(1 << 17)
    for n in list(range(0, 260)) + [rng.randrange(260, 1 << 17) for _ in range(300)] + [65280, 65536, (1 << 17) - 1]:
        off = rng.randrange(0, 7)
        n = min(n, len(blob) - off)
        piece = blob[off:off + n]
        assert L.np1_debug_crc32(piece, n) == (zlib.crc32(piece) & 0xffffffff), n


@pytest.fixture
def lane_decoder(monkeypatch):
    """the lane-per-block decoder of the device-side ingest (nextpolish_amd/csrc/np_inflate_lane.h: the same C++ the GPU lanes run)"""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "DECODER", "np1_debug_inflate_lane")


@pytest.mark.parametrize("level", [0, 1, 4, 6, 9])
def test_lane_decoder_matches_zlib_on_every_block_kind(level, lane_decoder):
    test_matches_zlib_on_every_block_kind(level)


def test_lane_decoder_multi_block_streams_and_damage(lane_decoder):
    test_multi_block_streams_and_flushes()
    test_damaged_or_mismatched_streams_are_never_wrong()


@pytest.fixture(params=["np1_debug_inflate_lds", "np1_debug_inflate_lds75", "np1_debug_inflate_lds64"])
def lds_decoder(request, monkeypatch):
    """the LDS-table lane decoder of the device-side ingest (nextpolish_amd/csrc/np_inflate_lds.h: the same C++ the GPU lanes run, over a plain
    array here) with 10 / 8-bit primary tables, with the 7 / 5-bit ones the kernel runs with by default, and with 6 / 4-bit ones, which send most codes down
    the canonical long-code path"""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "DECODER", request.param)


@pytest.mark.parametrize("level", [0, 1, 4, 6, 9])
def test_lds_decoder_matches_zlib_on_every_block_kind(level, lds_decoder):
    test_matches_zlib_on_every_block_kind(level)


def test_lds_decoder_multi_block_streams_and_damage(lds_decoder):
    test_multi_block_streams_and_flushes()
    test_damaged_or_mismatched_streams_are_never_wrong()


@pytest.mark.parametrize("decoder", ["host", "host-plain", "lane", "lds", "lds75", "lds64"])
def test_decoders_never_touch_a_byte_outside_their_buffers(decoder):
    """tests/model/inflate_fuzz.cpp under AddressSanitizer + UBSan: streams decoded from / into heap buffers of exactly their size, intact
    and damaged; includes the constructed case (1-4 literals, then a 258-byte far match ending 10-13 bytes before the end of the block) on
    which the host decoder used to write up to 4 bytes into the neighbouring block of a BGZF window (found in round 4)."""
    import os
    import subprocess
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model")
    b = subprocess.run(["make", "-C", d, "inflate_fuzz"], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-1500:]
    for seed in (1, 2, 3):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
        if decoder == "host-plain":      # the build of the loop without BMI2 (what a CPU without it runs)
            env["NP_INFLATE_PLAIN"] = "1"
        p = subprocess.run([os.path.join(d, "inflate_fuzz"), decoder.split("-")[0], "1200", str(seed)], capture_output=True, text=True, env=env)
        assert p.returncode == 0 and " 0 failures" in p.stdout, (p.stdout[-300:], p.stderr[-2500:])
