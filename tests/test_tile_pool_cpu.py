"""The tile -> workgroup assignment of the two tile walks (bloomscene_amd/csrc/common.h: pooled_tile, pooled_grid),
restated on the host: whatever order the pool draws are made in, every tile is rendered exactly once, the grid holds
enough workgroups, and the draw that resets the counter is the launch's last."""
import re
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _constants():
    src = open(os.path.join(ROOT, "bloomscene_amd", "csrc", "common.h")).read()
    k = int(re.search(r"#define BSR_POOL_K (\d+)", src).group(1))
    e = int(re.search(r"#define BSR_POOL_E (\d+)", src).group(1))
    m = re.search(r"return per >= (\d+) \* BSR_POOL_K \? BSR_POOL_K : 0;", src)
    assert m, "pool_tiles_per_band changed: restate it here"
    return k, e, int(m.group(1))


def _assignment(n_tiles, order_seed):
    K, E, FACTOR = _constants()
    per = (n_tiles + 7) >> 3
    k = K if per >= FACTOR * K else 0
    own = per - k
    grid = 8 * (per + (E if k else 0))
    rng = np.random.default_rng(order_seed)
    tiles, draws = [], 0
    pool_wgs = [b for b in range(grid) if (b >> 3) >= own]
    rng.shuffle(pool_wgs)   # pool workgroups reach their draw in any order
    for b in range(grid):
        if (b >> 3) < own:
            t = (b & 7) * per + (b >> 3)
            if t < n_tiles:
                tiles.append(t)
    reset_at = None
    for b in pool_wgs:
        j = draws
        draws += 1
        if j == 8 * (k + E) - 1:
            reset_at = draws
        t = (j & 7) * per + own + (j >> 3) if j < 8 * k else n_tiles
        if t < n_tiles:
            tiles.append(t)
    return tiles, grid, draws, reset_at, k


@pytest.mark.parametrize("n_tiles", [1, 7, 8, 9, 63, 1024, 2500, 4095, 4096, 4097, 8160, 8161, 8167, 16 * 8160, 65536, 70001])
def test_every_tile_is_assigned_exactly_once(n_tiles):
    for seed in (0, 1):
        tiles, grid, draws, reset_at, k = _assignment(n_tiles, seed)
        assert sorted(tiles) == list(range(n_tiles))
        assert grid >= n_tiles and grid % 8 == 0
        if k:
            assert reset_at == draws   # the counter is zeroed by the last draw of the launch, not before
        else:
            assert draws == 0


def test_small_launches_keep_the_static_order():
    K, E, FACTOR = _constants()
    assert _assignment(2500, 0)[4] == 0          # C2's 800 x 800: no pool
    assert _assignment(8160, 0)[4] == K          # 1920 x 1080: pooled
    assert _assignment(8 * FACTOR * K, 0)[4] == K and _assignment(8 * FACTOR * K - 8, 0)[4] == 0
