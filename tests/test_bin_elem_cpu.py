"""The 8-byte form of a binning element (bloomscene_amd/csrc/common.h: load_elem_m / store_elem_m / elem_key_m), restated
on the host: what a reader gets back is the tile's high byte (the low byte is the pass-1 bucket it reads from), the
Gaussian id and the depth bits; the sort key inside a tile is (depth bits, id) as in the 12-byte form."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_source_still_packs_the_way_this_test_restates_it():
    src = open(os.path.join(ROOT, "bloomscene_amd", "csrc", "common.h")).read()
    assert "make_uint2(((e.x >> 8) << 24) | e.y, e.z)" in src
    assert "BinElem{(v.x >> 24) << 8, v.x & 0x00ffffffu, v.y}" in src
    assert re.search(r"\(\(uint64_t\)v\.y << 32\) \| \(uint64_t\)\(v\.x & 0x00ffffffu\)", src)
    b = open(os.path.join(ROOT, "bloomscene_amd", "csrc", "binning.hip")).read()
    assert "const int compact = (tile_owned && P <= (1 << 24)) ? 1 : 0;" in b


def test_pack_unpack_round_trip_and_key_order():
    rng = np.random.default_rng(0)
    n = 200000
    tile = rng.integers(0, 1 << 16, n, dtype=np.uint64)
    gid = rng.integers(0, 1 << 24, n, dtype=np.uint64)
    gid[:4] = [0, (1 << 24) - 1, 1, (1 << 24) - 2]
    tile[:4] = [0, 65535, 255, 256]
    depth = rng.integers(0, 1 << 32, n, dtype=np.uint64)
    w0 = (((tile >> 8) << 24) | gid) & 0xffffffff
    w1 = depth
    # reader
    tile_back = (w0 >> 24) << 8
    id_back = w0 & 0x00ffffff
    assert np.array_equal(tile_back, tile & 0xff00)
    assert np.array_equal(id_back, gid)
    key = (w1 << 32) | id_back
    key12 = (depth << 32) | gid   # elem_key of the 12-byte form
    assert np.array_equal(key, key12)
    # the tile-owned pass uses (tile >> 8) & 255 of what it reads back
    assert np.array_equal((tile_back >> 8) & 255, (tile >> 8) & 255)
