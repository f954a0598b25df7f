"""Fused anchor expansion: the tail of BloomScene's ``generate_neural_gaussians``
(reference ``gaussian_renderer/__init__.py:165-203``), as one differentiable op on the MI355X.

The reference takes the per-anchor MLP outputs of the visible anchors, builds a ``[N*K, 22]``
concat, boolean-indexes it with ``neural_opacity > 0`` and post-processes the pieces with half a
dozen more torch kernels.  ``expand_anchors`` does the same arithmetic in two HIP kernels behind
the C ABI of ``include/bloomscene_anchors.h`` (selection count + fused gather/activate/scatter) and
one for the gradient.  Like everything in this package it has no CPU path.

    xyz, color, opacity, scaling, rot, mask = expand_anchors(
        anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot)

with the reference's names and shapes: ``anchor [N,3]``, ``grid_scaling [N,6]``, ``grid_offsets
[N,K,3]``, ``neural_opacity [N*K,1]`` (already multiplied by the binary grid mask, GR:168),
``color [N*K,3]``, ``scale_rot [N*K,7]``; outputs are what GR:174-201 produce, in the same order.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _capi


def _ptr(t):
    return None if t is None else t.data_ptr()


def _need_gpu(*tensors):
    for t in tensors:
        if t.device.type != "cuda":
            raise RuntimeError("bloomscene_amd.expand_anchors needs tensors on the GPU (there is no CPU path)")


def _f32c(t):
    return t.detach().contiguous().float()


class _ExpandAnchors(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot):
        _need_gpu(anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot)
        lib = _capi.lib()
        N = anchor.shape[0]
        K = grid_offsets.shape[1] if grid_offsets.dim() == 3 else (neural_opacity.numel() // max(N, 1))
        if anchor.dim() != 2 or anchor.shape[1] != 3:
            raise RuntimeError("anchor must have dimensions (num_anchors, 3)")
        if tuple(grid_scaling.shape) != (N, 6):
            raise RuntimeError("grid_scaling must have dimensions (num_anchors, 6)")
        if grid_offsets.numel() != N * K * 3 or neural_opacity.numel() != N * K or color.numel() != N * K * 3 \
                or scale_rot.numel() != N * K * 7:
            raise RuntimeError("per-candidate tensors must hold num_anchors * n_offsets rows")
        dev = anchor.device
        a, gs, go, no, co, sr = (_f32c(t) for t in (anchor, grid_scaling, grid_offsets, neural_opacity, color,
                                                    scale_rot))
        stream = torch.cuda.current_stream(dev).cuda_stream
        mask = torch.empty(N * K, dtype=torch.bool, device=dev)
        if N * K == 0:
            S = 0
            scratch = torch.empty(0, dtype=torch.uint8, device=dev)
        else:
            nbytes = lib.bsr_anchor_scratch_bytes(N, K)
            if nbytes == 0:
                raise RuntimeError("expand_anchors: n_offsets must be in 1..256 and N*K < 2^31")
            scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            count = C.c_int(0)
            _capi.check(lib.bsr_anchor_select(N, K, _ptr(no), _ptr(mask), _ptr(scratch), C.byref(count), stream),
                        "bsr_anchor_select")
            S = count.value
        xyz = torch.empty(S, 3, dtype=torch.float32, device=dev)
        color_out = torch.empty(S, 3, dtype=torch.float32, device=dev)
        opacity = torch.empty(S, 1, dtype=torch.float32, device=dev)
        scaling = torch.empty(S, 3, dtype=torch.float32, device=dev)
        rot = torch.empty(S, 4, dtype=torch.float32, device=dev)
        if S:
            _capi.check(lib.bsr_anchor_expand(N, K, S, _ptr(a), _ptr(gs), _ptr(go), _ptr(no), _ptr(co), _ptr(sr),
                                              _ptr(scratch), _ptr(xyz), _ptr(color_out), _ptr(opacity), _ptr(scaling),
                                              _ptr(rot), stream), "bsr_anchor_expand")
        ctx.dims = (N, K, S)
        ctx.shapes = tuple(t.shape for t in (anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot))
        ctx.save_for_backward(gs, go, no, sr, scratch)
        ctx.mark_non_differentiable(mask)
        return xyz, color_out, opacity, scaling, rot, mask

    @staticmethod
    def backward(ctx, g_xyz, g_color, g_opacity, g_scaling, g_rot, _g_mask):
        N, K, S = ctx.dims
        gs, go, no, sr, scratch = ctx.saved_tensors
        dev = gs.device
        lib = _capi.lib()

        def up(g):   # a missing upstream gradient means zeros
            return None if g is None or S == 0 else g.contiguous().float()
        g_xyz, g_color, g_opacity, g_scaling, g_rot = (up(g) for g in (g_xyz, g_color, g_opacity, g_scaling, g_rot))
        d_anchor = torch.empty(N, 3, dtype=torch.float32, device=dev)
        d_gs = torch.empty(N, 6, dtype=torch.float32, device=dev)
        d_go = torch.empty(N * K, 3, dtype=torch.float32, device=dev)
        d_no = torch.empty(N * K, dtype=torch.float32, device=dev)
        d_co = torch.empty(N * K, 3, dtype=torch.float32, device=dev)
        d_sr = torch.empty(N * K, 7, dtype=torch.float32, device=dev)
        if N * K:
            stream = torch.cuda.current_stream(dev).cuda_stream
            _capi.check(lib.bsr_anchor_expand_backward(
                N, K, S, _ptr(gs), _ptr(go), _ptr(no), _ptr(sr), _ptr(scratch), _ptr(g_xyz), _ptr(g_color),
                _ptr(g_opacity), _ptr(g_scaling), _ptr(g_rot), _ptr(d_anchor), _ptr(d_gs), _ptr(d_go), _ptr(d_no),
                _ptr(d_co), _ptr(d_sr), stream), "bsr_anchor_expand_backward")
        sh = ctx.shapes
        return (d_anchor.view(sh[0]), d_gs.view(sh[1]), d_go.view(sh[2]), d_no.view(sh[3]), d_co.view(sh[4]),
                d_sr.view(sh[5]))


def expand_anchors(anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot):
    """-> (xyz [S,3], color [S,3], opacity [S,1], scaling [S,3], rot [S,4], mask [N*K] bool).
    Replaces reference gaussian_renderer/__init__.py:169-201 (see the module docstring)."""
    return _ExpandAnchors.apply(anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot)
