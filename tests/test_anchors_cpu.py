"""CPU tests of the anchor-expansion oracle (oracle/anchors.py) and of the host side's error
behaviour.  The oracle restates reference gaussian_renderer/__init__.py:165-203 with torch ops;
here it is checked against an independent scalar re-derivation of the same lines."""
import math

import numpy as np
import pytest
import torch

from oracle import anchors as OA


def _scalar_model(anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot):
    """Candidate by candidate, in python floats (binary64)."""
    N, K = grid_offsets.shape[:2]
    rows = []
    mask = []
    for n in range(N):
        for k in range(K):
            i = n * K + k
            op = float(neural_opacity[i, 0])
            mask.append(op > 0.0)
            if not op > 0.0:
                continue
            sr = [float(v) for v in scale_rot[i]]
            gs = [float(v) for v in grid_scaling[n]]
            scaling = [gs[3 + c] / (1.0 + math.exp(-sr[c])) for c in range(3)]
            nrm = max(math.sqrt(sum(v * v for v in sr[3:7])), 1e-12)
            rot = [v / nrm for v in sr[3:7]]
            xyz = [float(anchor[n, c]) + float(grid_offsets[n, k, c]) * gs[c] for c in range(3)]
            rows.append(xyz + [float(v) for v in color[i]] + [op] + scaling + rot)
    return np.array(rows, dtype=np.float64).reshape(-1, 14), np.array(mask)


@pytest.mark.parametrize("N,K,seed", [(37, 10, 0), (8, 5, 1), (50, 1, 2), (3, 256, 3)])
def test_oracle_matches_scalar_model(N, K, seed):
    inp = OA.synthetic_anchor_inputs(N, K, seed=seed, dtype=torch.float64, zero_quat_rows=2)
    xyz, col, op, sc, rot, mask = OA.expand_anchors_reference(*inp)
    rows, m = _scalar_model(*inp)
    assert np.array_equal(mask.numpy(), m)
    got = torch.cat([xyz, col, op, sc, rot], dim=1).numpy()
    assert got.shape == rows.shape
    np.testing.assert_allclose(got, rows, rtol=1e-13, atol=1e-300)
    assert 0 < mask.sum() < mask.numel()


def test_oracle_autograd_sums_over_offsets():
    """d/d anchor of sum(xyz) = number of selected offsets of that anchor (GR:188,201)."""
    N, K = 20, 10
    inp = [t.requires_grad_(True) for t in OA.synthetic_anchor_inputs(N, K, seed=4, dtype=torch.float64)]
    xyz, col, op, sc, rot, mask = OA.expand_anchors_reference(*inp)
    xyz.sum().backward()
    per_anchor = mask.view(N, K).sum(1).double()
    assert torch.equal(inp[0].grad, per_anchor[:, None].expand(N, 3))
    assert torch.equal(inp[2].grad.view(N * K, 3)[~mask], torch.zeros_like(inp[2].grad.view(N * K, 3)[~mask]))


def test_product_has_no_cpu_path():
    from bloomscene_amd.neural_gaussians import expand_anchors
    inp = OA.synthetic_anchor_inputs(4, 10, seed=0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        expand_anchors(*inp)
