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
        # (the host was blocked on the count: the GPU idles until the expansion is launched, so ONE allocation,
        # carved into the five outputs -- rows of 14 floats would interleave them, the sections keep each contiguous)
        packed = torch.empty(S * 14, dtype=torch.float32, device=dev)
        rot = packed[:4 * S].view(S, 4)                    # first: its rows are read as 16-byte vectors
        xyz = packed[4 * S:7 * S].view(S, 3)
        color_out = packed[7 * S:10 * S].view(S, 3)
        scaling = packed[10 * S:13 * S].view(S, 3)
        opacity = packed[13 * S:14 * S].view(S, 1)
        if S:
            _capi.check(lib.bsr_anchor_expand(N, K, S, _ptr(a), _ptr(gs), _ptr(go), _ptr(no), _ptr(co), _ptr(sr),
                                              _ptr(scratch), _ptr(xyz), _ptr(color_out), _ptr(opacity), _ptr(scaling),
                                              _ptr(rot), stream), "bsr_anchor_expand")
        # the mask is an index output and unused outputs need no zero gradients: without this autograd fills a bool
        # [N*K] zero tensor for the mask's "gradient" on every backward (26 us at 1 M candidates)
        ctx.set_materialize_grads(False)
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


class _NeuralRender(torch.autograd.Function):
    """expand_anchors + the rasterizer call of gaussian_renderer.render (GR:165-203, 235-262) as ONE autograd node over
    ONE native call each way (include/bloomscene_anchors.h: bsr_anchor_render_forward / _backward).  The host is
    blocked on the number of selected Gaussians; done from python as two Functions, the interpreter's work between the
    count and the rasterizer's first launch is GPU idle time (100 of 630 us per step at BloomScene's
    shape).  Outputs and gradients are bit-identical to the two separate Functions."""

    @staticmethod
    def forward(ctx, anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot, raster_settings,
                depth_gradient, flags=0, capacity=None):
        from .rasterizer import _Scratch, _dev_f32
        _need_gpu(anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot)
        lib = _capi.lib()
        rs = raster_settings
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
        H, W = int(rs.image_height), int(rs.image_width)
        a, gs, go, no, co, sr = (_f32c(t) for t in (anchor, grid_scaling, grid_offsets, neural_opacity, color,
                                                    scale_rot))
        bg, view, proj, campos = (_dev_f32(t, n, dev) for t, n in ((rs.bg, "bg"), (rs.viewmatrix, "viewmatrix"),
                                                                    (rs.projmatrix, "projmatrix"), (rs.campos, "campos")))
        stream = torch.cuda.current_stream(dev).cuda_stream
        mask = torch.empty(N * K, dtype=torch.bool, device=dev)
        nbytes = lib.bsr_anchor_scratch_bytes(N, K) if N * K else 0
        if N * K and nbytes == 0:
            raise RuntimeError("expand_anchors: n_offsets must be in 1..256 and N*K < 2^31")
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out_color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        packed_box = []

        def _alloc_gaussians(_user, n):
            t = torch.empty(int(n) // 4, dtype=torch.float32, device=dev)
            packed_box.append(t)
            return t.data_ptr()

        gauss_cb = _capi.ALLOC_FN(_alloc_gaussians)
        geom, binning, img = _Scratch(dev), _Scratch(dev), _Scratch(dev)
        # capacity: BSR_FLAG_NO_READBACK -- static shapes (every output has N * K rows, the unselected tail padded with
        # invisible Gaussians), no host wait for the selection count or for num_rendered
        if capacity is not None:
            if int(capacity) <= 0 or N * K == 0:
                raise RuntimeError("capacity mode needs a positive capacity and at least one candidate")
            flags = int(flags) | 4
        S, R = C.c_int(0), C.c_int(int(capacity) if capacity is not None else 0)
        with torch.cuda.device(dev):
            rc = lib.bsr_anchor_render_forward(
                N, K, _ptr(a), _ptr(gs), _ptr(go), _ptr(no), _ptr(co), _ptr(sr), _ptr(mask), _ptr(scratch),
                gauss_cb, None, geom.callback, None, binning.callback, None, img.callback, None,
                _ptr(bg), W, H, float(rs.scale_modifier), _ptr(view), _ptr(proj), _ptr(campos), float(rs.tanfovx),
                float(rs.tanfovy), out_color.data_ptr(), out_depth.data_ptr(), int(bool(rs.debug)), stream,
                C.byref(S), C.byref(R), int(flags))
        _capi.check(rc, "bsr_anchor_render_forward")
        S, R = S.value, R.value
        # (the last buffer asked for: the first request may have been a guess made before the count was known)
        packed = packed_box[-1] if packed_box else torch.empty(0, dtype=torch.float32, device=dev)
        rot = packed[:4 * S].view(S, 4)
        xyz = packed[4 * S:7 * S].view(S, 3)
        rgb = packed[7 * S:10 * S].view(S, 3)
        scaling = packed[10 * S:13 * S].view(S, 3)
        opacity = packed[13 * S:14 * S].view(S, 1)
        radii = packed[14 * S:15 * S].view(torch.int32)
        # GR:224-229: a zeros tensor whose .grad receives the screen-space gradient; its VALUE is never read
        viewspace = torch.zeros((S, 3), dtype=torch.float32, device=dev)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(radii, mask, viewspace)
        ctx.dims = (N, K, S, R, H, W)
        ctx.flags = int(flags)
        ctx.raster_settings = rs
        ctx.shapes = tuple(t.shape for t in (anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot))
        ctx.cam = (bg, view, proj, campos)
        ctx.viewspace = viewspace
        ctx.out_depth = out_depth.detach().clone() if depth_gradient else None
        ctx.save_for_backward(gs, go, no, sr, scratch, packed, geom.tensor, binning.tensor, img.tensor)
        return out_color, out_depth, radii, mask, xyz, rgb, opacity, scaling, rot, viewspace

    @staticmethod
    def backward(ctx, g_image, g_depth, _g_radii, _g_mask, g_xyz, g_rgb, g_opacity, g_scaling, g_rot, _g_view):
        N, K, S, R, H, W = ctx.dims
        rs = ctx.raster_settings
        gs, go, no, sr, scratch, packed, geom_t, bin_t, img_t = ctx.saved_tensors
        bg, view, proj, campos = ctx.cam
        dev = gs.device
        lib = _capi.lib()
        if g_image is None:   # only the depth output was used downstream
            g_image = torch.zeros((3, H, W), dtype=torch.float32, device=dev)
        g_image = g_image.contiguous().float()
        if ctx.out_depth is not None and g_depth is None:
            g_depth = torch.zeros((1, H, W), dtype=torch.float32, device=dev)
        g_depth = None if g_depth is None else g_depth.contiguous().float()

        def up(g):
            return None if g is None or S == 0 else g.contiguous().float()
        g_xyz, g_rgb, g_opacity, g_scaling, g_rot = (up(g) for g in (g_xyz, g_rgb, g_opacity, g_scaling, g_rot))
        grads = torch.empty(17 * S, dtype=torch.float32, device=dev)
        d_anchor = torch.empty(N, 3, dtype=torch.float32, device=dev)
        d_gs = torch.empty(N, 6, dtype=torch.float32, device=dev)
        d_go = torch.empty(N * K, 3, dtype=torch.float32, device=dev)
        d_no = torch.empty(N * K, dtype=torch.float32, device=dev)
        d_co = torch.empty(N * K, 3, dtype=torch.float32, device=dev)
        d_sr = torch.empty(N * K, 7, dtype=torch.float32, device=dev)
        if N * K:
            stream = torch.cuda.current_stream(dev).cuda_stream
            with torch.cuda.device(dev):
                rc = lib.bsr_anchor_render_backward(
                    N, K, S, R, _ptr(gs), _ptr(go), _ptr(no), _ptr(sr), _ptr(scratch), _ptr(packed) if S else None,
                    geom_t.data_ptr() if geom_t.numel() else None, bin_t.data_ptr() if bin_t.numel() else None,
                    img_t.data_ptr() if img_t.numel() else None, _ptr(bg), W, H, float(rs.scale_modifier), _ptr(view),
                    _ptr(proj), _ptr(campos), float(rs.tanfovx), float(rs.tanfovy), _ptr(g_image), _ptr(ctx.out_depth),
                    _ptr(g_depth), _ptr(g_xyz), _ptr(g_rgb), _ptr(g_opacity), _ptr(g_scaling), _ptr(g_rot),
                    _ptr(grads) if S else None, _ptr(d_anchor), _ptr(d_gs), _ptr(d_go), _ptr(d_no), _ptr(d_co), _ptr(d_sr),
                    int(bool(rs.debug)), stream, ctx.flags)
            _capi.check(rc, "bsr_anchor_render_backward")
        # the screen-space gradient, where the reference's consumers look for it (viewspace_points.grad)
        ctx.viewspace.grad = grads[14 * S:17 * S].view(S, 3)
        sh = ctx.shapes
        return (d_anchor.view(sh[0]), d_gs.view(sh[1]), d_go.view(sh[2]), d_no.view(sh[3]), d_co.view(sh[4]),
                d_sr.view(sh[5]), None, None, None, None)


def render_anchors(anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot, raster_settings,
                   depth_gradient=False, flags=None, capacity=None):
    """-> (image [3,H,W], depth [1,H,W], radii int32 [S], mask bool [N*K], xyz, color, opacity, scaling, rot,
    viewspace_points [S,3]) -- expand_anchors followed by GaussianRasterizer(raster_settings)(colors_precomp=color, ...)
    in one native call each way; after backward ``viewspace_points.grad`` holds the screen-space gradient.
    ``flags``: BSR_FLAG_* of the call; None = the calling thread's ``numerics(...)`` context.
    ``capacity`` (tile instances; extension): STATIC SHAPES and no host wait (include/bloomscene_anchors.h,
    BSR_FLAG_NO_READBACK) -- every per-Gaussian output then has N * K rows: the S selected Gaussians first, in the usual
    order, then padding rows no camera sees (radii 0, zero gradients); S = mask.sum() stays on the device.  The image,
    the depth and the gradients of the six inputs are bit-identical to the default mode's; a warmed-up forward +
    backward can be captured into a HIP graph."""
    from .numerics import resolve_flags
    return _NeuralRender.apply(anchor, grid_scaling, grid_offsets, neural_opacity, color, scale_rot, raster_settings,
                               bool(depth_gradient), resolve_flags() if flags is None else int(flags), capacity)
