"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the anchor expansion.

A torch restatement, op for op, of the tail of the reference's ``generate_neural_gaussians``
(``/root/reference/gaussian_renderer/__init__.py:165-203``; ``rotation_activation`` is
``torch.nn.functional.normalize``, ``scene/gaussian_model.py:121``; einops' ``'n (c) -> (n k)
(c)'`` repeat is ``repeat_interleave`` on dim 0).  It runs on CPU tensors in whatever dtype it is
given (float64 for gradient checks) and torch.autograd provides the gradient oracle -- in the
reference, too, the backward of these lines is whatever autograd derives.

PARITY UNPINNED: the reference module cannot be imported here (its imports need torch_scatter,
plyfile and the unbuilt CUDA extension) and its tests hold no vectors for these lines, so nothing
pins this restatement beyond the cited source.  Only tests/ and tools/bench_anchors.py's CPU leg
may import this file; the product (bloomscene_amd.neural_gaussians) never does.
"""
from __future__ import annotations

import torch


def expand_anchors_reference(anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot):
    """-> (xyz, color, opacity, scaling, rot, mask) exactly as GR:169-201 computes them."""
    n_anchor = anchor.shape[0]
    n_offsets = grid_offsets.shape[1]
    neural_opacity = neural_opacity.reshape([-1, 1])                      # GR:167
    mask = (neural_opacity > 0.0).view(-1)                                 # GR:169-170
    opacity = neural_opacity[mask]                                         # GR:174
    color = color.reshape([n_anchor * n_offsets, 3])                       # GR:178
    scale_rot = scale_rot.reshape([n_anchor * n_offsets, 7])               # GR:182
    offsets = grid_offsets.reshape([-1, 3])                                # GR:184
    per_anchor = torch.cat([grid_scaling, anchor], dim=-1)                 # GR:187  [N, 6+3]
    per_candidate = per_anchor.repeat_interleave(n_offsets, dim=0)         # GR:188  [N*K, 9]
    everything = torch.cat([per_candidate, color, scale_rot, offsets], dim=-1)   # GR:189-190
    chosen = everything[mask]                                              # GR:192
    scaling_repeat, repeat_anchor, color, scale_rot, offsets = chosen.split([6, 3, 3, 7, 3], dim=-1)   # GR:193
    scaling = scaling_repeat[:, 3:] * torch.sigmoid(scale_rot[:, :3])      # GR:196-197
    rot = torch.nn.functional.normalize(scale_rot[:, 3:7])                 # GR:198
    offsets = offsets * scaling_repeat[:, :3]                              # GR:200
    xyz = repeat_anchor + offsets                                          # GR:201
    return xyz, color, opacity, scaling, rot, mask


def synthetic_anchor_inputs(n_anchor, n_offsets, seed=0, dtype=torch.float32, keep_fraction=0.5, zero_quat_rows=0):
    """Seeded inputs with the reference's shapes and plausible magnitudes: anchors in a 10 m box,
    grid_scaling = exp-activated voxel scales, offsets ~ N(0, 1), MLP heads ~ N(0, 1); roughly
    ``keep_fraction`` of the candidates have positive opacity, the rest are <= 0 (incl. exact 0s,
    which is what the binary grid mask of GR:168 produces)."""
    g = torch.Generator().manual_seed(seed)
    n = n_anchor * n_offsets
    anchor = (torch.rand(n_anchor, 3, generator=g) * 10 - 5).to(dtype)
    grid_scaling = torch.exp(torch.randn(n_anchor, 6, generator=g) * 0.5 - 3.0).to(dtype)
    grid_offsets = torch.randn(n_anchor, n_offsets, 3, generator=g).to(dtype)
    neural_opacity = torch.tanh(torch.randn(n, 1, generator=g))
    u = torch.rand(n, 1, generator=g)
    neural_opacity = torch.where(u < keep_fraction, neural_opacity.abs() + 1e-3, -neural_opacity.abs())
    neural_opacity = torch.where(u > 0.9, torch.zeros_like(neural_opacity), neural_opacity).to(dtype)
    color = torch.rand(n, 3, generator=g).to(dtype)
    scale_rot = (torch.randn(n, 7, generator=g) * 1.5).to(dtype)
    if zero_quat_rows:
        scale_rot[torch.randperm(n, generator=g)[:zero_quat_rows], 3:7] = 0.0   # exercises normalize's eps
    return anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot
