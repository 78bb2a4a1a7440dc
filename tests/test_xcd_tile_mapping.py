"""xcd_tile() (blazeseq_amd/csrc/bzq_device.hpp): workgroup b -> tile, one contiguous range of tiles per XCD (b % 8).  The
formula restated here must be a bijection onto [0, grid) for every grid size, hand the tiles of one XCD out in order, and
keep the ranges of the eight XCDs contiguous and disjoint -- and the kernel source must still hold the same formula."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def xcd_tile(b: int, grid: int) -> int:
    q, r, x = grid >> 3, grid & 7, b & 7
    return x * q + min(x, r) + (b >> 3)


def test_bijection_and_contiguity():
    for grid in list(range(1, 200)) + [1023, 1024, 1025, 4093, 194091, 1525879]:
        tiles = [xcd_tile(b, grid) for b in range(grid)] if grid < 5000 else None
        if tiles is not None:
            assert sorted(tiles) == list(range(grid)), grid
        for x in range(8):   # the tiles of XCD x: consecutive workgroups b = x, x + 8, ... take consecutive tiles
            bs = range(x, grid, 8)
            if len(bs) == 0:
                continue
            first = xcd_tile(bs[0], grid)
            step = max(1, len(bs) // 50)
            for i in range(0, len(bs), step):
                assert xcd_tile(bs[i], grid) == first + i
            if x < 7 and len(range(x + 1, grid, 8)):
                assert xcd_tile(x + 1, grid) == first + len(bs)   # the next XCD starts where this one ends


def test_source_holds_the_same_formula():
    src = open(os.path.join(ROOT, "blazeseq_amd", "csrc", "bzq_device.hpp")).read()
    body = re.search(r"int64_t xcd_tile\(\) \{(.*?)\n\}", src, re.S).group(1)
    assert "gridDim.x >> 3" in body and "gridDim.x & 7u" in body and "blockIdx.x & 7u" in body
    assert "x * q" in body and "(x < r ? x : r)" in body and "blockIdx.x >> 3" in body
