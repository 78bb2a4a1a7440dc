"""bzq_bgzf_inflate (blazeseq_amd/csrc/bzq_inflate.hpp): BGZF blocks inflated by the GPU == the bytes zlib compressed, over
every DEFLATE block type (stored, fixed, dynamic), strategies that stress the copy (RLE: distance 1, overlapping), sizes from
0 to 64 KiB, and the FASTQ the ingest sees."""
import ctypes as C
import zlib

import numpy as np
import pytest

from blazeseq_amd import _lib as L
from blazeseq_amd.parser import Context
from tests.bgzf_util import bgzf_block, bgzf_compress
from tests.fastq_fuzz import rand_stream

pytestmark = pytest.mark.gpu


# the product kernel (one wave per block, bzq_inflate.hpp) and -- in the EXPERIMENTS library only (tests/test_gpu_experiments.py runs
# this file against it with BZQ_TEST_EXPERIMENTS=1) -- the round-4 one that decodes eight blocks per wave
# (experiments/csrc/bzq_inflate_ms.hpp, option inflate_ms: correct, 2.7x slower, profiles/r4_inflate_ms.md)
import os
EXPERIMENTS = os.environ.get("BZQ_TEST_EXPERIMENTS") == "1"
MS_KERNELS, MS_IDS = ([0, 1], ["wave_per_block", "eight_blocks_per_wave"]) if EXPERIMENTS else ([0], ["wave_per_block"])


def pick_kernel(ctx, ms):
    if ms:
        ctx.set_option("inflate_ms", 1)

def inflate_on_device(ctx, comp: bytes) -> bytes:
    a = np.frombuffer(comp, dtype=np.uint8)
    blocks, n, consumed, out_bytes = ctx.bgzf_scan(a)
    assert consumed == a.size
    d_c, d_o = C.c_void_p(), C.c_void_p()
    assert L.lib().bzq_device_alloc(ctx.h, a.size + 64, C.byref(d_c)) == 0
    assert L.lib().bzq_device_alloc(ctx.h, out_bytes + 64, C.byref(d_o)) == 0
    try:
        assert L.lib().bzq_copy_to_device(ctx.h, d_c, a.ctypes.data, a.size) == 0
        ctx.bgzf_inflate(d_c.value, a.size, blocks, n, d_o.value, out_bytes)
        out = np.empty(out_bytes, dtype=np.uint8)
        if out_bytes:
            assert L.lib().bzq_copy_to_host(ctx.h, out.ctypes.data, d_o, out_bytes) == 0
        return out.tobytes()
    finally:
        L.lib().bzq_device_free(ctx.h, d_c); L.lib().bzq_device_free(ctx.h, d_o)


def payloads():
    rng = np.random.default_rng(11)
    fq = rand_stream(rng, n_records=3000, max_len=150, dirty=0.0, tail=0)
    yield "fastq", fq
    yield "empty", b""
    yield "one_byte", b"A"
    yield "same_byte", b"G" * 200000
    yield "period3", b"ACG" * 70000
    yield "random", rng.integers(0, 256, 150000, dtype=np.uint8).tobytes()
    yield "low_entropy", rng.integers(65, 69, 300000, dtype=np.uint8).tobytes()
    yield "text", (b"the quick brown fox jumps over the lazy dog\n" * 5000)[:180001]
    yield "long_matches", (bytes(rng.integers(0, 256, 300, dtype=np.uint8)) * 800)
    yield "far_matches", (bytes(rng.integers(0, 256, 31000, dtype=np.uint8)) * 6)


@pytest.mark.parametrize("ms", MS_KERNELS, ids=MS_IDS)
@pytest.mark.parametrize("name,data", list(payloads()), ids=[n for n, _ in payloads()])
def test_levels_and_strategies(name, data, ms):
    ctx = Context()
    pick_kernel(ctx, ms)
    for level, strategy in [(6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (0, zlib.Z_DEFAULT_STRATEGY),
                            (6, zlib.Z_FIXED), (6, zlib.Z_RLE), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_FILTERED)]:
        block = 65280 if level else 60000
        if name == "random" and level:
            block = 32000   # incompressible bytes grow a little
        comp = bgzf_compress(data, block=block, level=level, strategy=strategy)
        got = inflate_on_device(ctx, comp)
        assert got == data, (name, level, strategy, len(got), len(data))


@pytest.mark.parametrize("ms", MS_KERNELS, ids=MS_IDS)
def test_block_sizes_around_the_edges(ms):
    ctx = Context()
    pick_kernel(ctx, ms)
    rng = np.random.default_rng(5)
    base = rand_stream(rng, n_records=700, max_len=150, dirty=0.0, tail=0)
    parts, want = [], []
    for n in [0, 1, 2, 3, 63, 64, 65, 255, 256, 257, 258, 259, 4095, 4096, 4097, 65279, 65280, 65535, 65536]:
        piece = (base * (n // len(base) + 1))[:n]
        parts.append(bgzf_block(piece)); want.append(piece)
    assert inflate_on_device(ctx, b"".join(parts)) == b"".join(want)


def test_the_last_symbols_of_many_blocks():
    """Every block's last two bytes are decoded outside the hand-written loop (it leaves with 'fewer than two bytes left'),
    from whatever the bit buffer holds at that moment: hundreds of blocks of ordinary FASTQ at the levels whose codes are
    longest (a level-1 block of the benchmark's data once failed exactly there -- the loop had left in front of its refill)."""
    from oracle import oracle as O
    ctx = Context()
    data = O.generate_synthetic(60000, 150, 150, 33, 73, "generic").tobytes()   # ~19 MB: ~290 blocks
    for level in (1, 2, 9):
        comp = bgzf_compress(data, block=65280, level=level)
        assert inflate_on_device(ctx, comp) == data, level


@pytest.mark.parametrize("ms", MS_KERNELS, ids=MS_IDS)
def test_what_a_sequencer_writes(ms):
    """Quality runs, duplicate reads, poly-G tails (tests/gzip_util.py): overlapping and long matches, taken in pieces by the
    symbol loop."""
    from tests.gzip_util import sequencer_like
    ctx = Context()
    data, _ = sequencer_like(5 << 20)
    for level in (1, 6, 9):
        assert inflate_on_device(ctx, bgzf_compress(data, block=65280, level=level)) == data, level


@pytest.mark.parametrize("ms", MS_KERNELS, ids=MS_IDS)
def test_many_blocks_random_mix(ms):
    ctx = Context()
    pick_kernel(ctx, ms)
    rng = np.random.default_rng(8)
    parts, want = [], []
    for i in range(600):
        kind = int(rng.integers(0, 5))
        n = int(rng.integers(0, 65000))
        if kind == 0:
            piece = rand_stream(rng, n_records=max(1, n // 330), max_len=150, dirty=0.0, tail=0)[:n]
        elif kind == 1:
            piece = rng.integers(0, 256, min(n, 30000), dtype=np.uint8).tobytes()
        elif kind == 2:
            piece = bytes([int(rng.integers(0, 256))]) * n
        elif kind == 3:
            piece = (bytes(rng.integers(0, 4, int(rng.integers(1, 40)), dtype=np.uint8) + 65) * 70000)[:n]
        else:
            piece = rng.integers(0, int(rng.integers(2, 200)), n, dtype=np.uint8).tobytes()
        level = int(rng.choice([0, 1, 4, 6, 9]))
        strat = int(rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY]))
        parts.append(bgzf_block(piece, level, strat)); want.append(piece)
    assert inflate_on_device(ctx, b"".join(parts)) == b"".join(want)


@pytest.mark.parametrize("ms", MS_KERNELS, ids=MS_IDS)
def test_corrupt_blocks_fail(ms):
    """A damaged payload fails -- a bad code, distance, length or size while decoding, or the CRC-32 of the output afterwards
    (RFC 1952 8.) -- unless the damage sits in bits the stream does not use; never a crash, a hang, a write outside the
    block's output or different bytes delivered."""
    ctx = Context()
    pick_kernel(ctx, ms)
    rng = np.random.default_rng(2)
    data = rand_stream(rng, n_records=190, max_len=150, dirty=0.0, tail=0)
    good = bgzf_block(data)
    guard = bgzf_block(b"Z" * 1000)
    failures = 0
    for trial in range(80):
        bad = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            bad[int(rng.integers(18, len(good) - 8))] ^= 1 << int(rng.integers(0, 8))
        try:
            out = inflate_on_device(ctx, bytes(bad) + guard)
            assert out == data + b"Z" * 1000   # the flipped bits were padding
        except RuntimeError as e:
            assert "block 0 failed to inflate" in str(e)
            failures += 1
    assert failures >= 75
    for name, off in (("crc", -8), ("isize", -4)):
        wrong = bytearray(good)
        wrong[off] ^= 0x01
        with pytest.raises(RuntimeError, match="block 0 failed"):
            inflate_on_device(ctx, bytes(wrong))
    second = bytearray(good + good)   # the verdict names the first failing block
    second[len(good) + 100] ^= 0x10
    with pytest.raises(RuntimeError, match="block 1 failed"):
        inflate_on_device(ctx, bytes(second))


@pytest.mark.parametrize("ms", MS_KERNELS, ids=MS_IDS)
def test_crc_of_every_size_class(ms):
    """The CRC pieces: 64 lanes x ceil(n / 64) bytes, the last lanes empty or short."""
    ctx = Context()
    pick_kernel(ctx, ms)
    rng = np.random.default_rng(4)
    parts, want = [], []
    for n in list(range(0, 200)) + [255, 256, 257, 4095, 4096, 4097, 65535, 65536]:
        piece = rng.integers(0, 256, n, dtype=np.uint8).tobytes() if n < 30000 else (b"ACGT" * 16384)[:n]
        parts.append(bgzf_block(piece, 1)); want.append(piece)
    assert inflate_on_device(ctx, b"".join(parts)) == b"".join(want)
