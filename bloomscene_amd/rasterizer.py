"""Host-side mirror of the reference's ``depth_diff_gaussian_rasterization`` python module.

Same names, argument meaning, return order and error behaviour as
``/root/reference/submodules/depth-diff-gaussian-rasterization/depth_diff_gaussian_rasterization/__init__.py``
(cited below as PYW) plus the torch glue of ``rasterize_points.cu`` (RP), but every native call
goes through the C ABI of ``include/bloomscene_rast.h`` into hand-written gfx950 kernels.

There is no CPU path here: tensors must live on the GPU and the HIP library must be present.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _capi
from .numerics import FLAG_EXACT_EXP, FLAG_EXACT_GRAD, numerics, resolve_flags  # noqa: F401  (re-exported)

FLAG_NO_READBACK = 4   # BSR_FLAG_NO_READBACK (include/bloomscene_rast.h)


def cpu_deep_copy_tuple(input_tuple):  # PYW:17-19
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


class GaussianRasterizationSettings(NamedTuple):  # PYW:158-170 (same 12 fields, same order)
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ------------------------------------------------------------------ pointer plumbing
def _dev_f32(t: torch.Tensor, name: str, device: torch.device):
    """contiguous fp32 tensor on `device`, or None for an absent (empty) optional input.
    RP:95-113 calls .contiguous().data_ptr<float>() on everything; an empty CPU tensor is the
    reference's encoding of nullptr (PYW:198-208)."""
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        raise RuntimeError(f"{name} must be on {device} (got {t.device}); bloomscene_amd has no CPU path")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream_handle(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _Scratch:
    """One growable byte tensor + the C callback that resizes it: the counterpart of
    resizeFunctional (RP:27-33)."""

    def __init__(self, device):
        tensor = torch.empty(0, dtype=torch.uint8, device=device)
        self.tensor = tensor

        # the closure must not capture `self`: self -> callback -> closure -> self would be a reference cycle that
        # keeps the scratch tensors (hundreds of MB per call) alive until the cycle collector runs
        def _resize(_user, nbytes):
            tensor.resize_(int(nbytes))
            return tensor.data_ptr()

        self.callback = _capi.ALLOC_FN(_resize)


def _check_means3D(means3D):
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")  # RP:57-59


# ------------------------------------------------------------------ native entry points
def _rasterize_gaussians_native(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                                campos, prefiltered, debug, flags=None, capacity=None):
    """Counterpart of `_C.rasterize_gaussians` = RasterizeGaussiansCUDA (RP:35-117).  ``flags``: the call's numerics
    (BSR_FLAG_* of include/bloomscene_rast.h); None = the calling thread's `numerics` context (default 0, the fast path).
    ``capacity`` (tile instances): BSR_FLAG_NO_READBACK -- the caller sizes the binning scratch, the call never waits
    for the GPU and the returned num_rendered IS the capacity (what the backward must be handed)."""
    flags = resolve_flags() if flags is None else int(flags)
    if capacity is not None:
        if int(capacity) <= 0:
            raise RuntimeError("capacity must be a positive number of tile instances")
        flags |= FLAG_NO_READBACK
    _check_means3D(means3D)
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor; bloomscene_amd has no CPU path")
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    out_color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    out_depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    geom, binning, img = _Scratch(dev), _Scratch(dev), _Scratch(dev)
    M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0  # RP:84-88
    t = dict(bg=_dev_f32(bg, "bg", dev), means3D=_dev_f32(means3D, "means3D", dev),
             colors=_dev_f32(colors, "colors_precomp", dev), opacity=_dev_f32(opacity, "opacities", dev),
             scales=_dev_f32(scales, "scales", dev), rotations=_dev_f32(rotations, "rotations", dev),
             cov3D=_dev_f32(cov3D_precomp, "cov3D_precomp", dev), view=_dev_f32(viewmatrix, "viewmatrix", dev),
             proj=_dev_f32(projmatrix, "projmatrix", dev), sh=_dev_f32(sh, "shs", dev),
             campos=_dev_f32(campos, "campos", dev))
    num_rendered = C.c_int(int(capacity) if capacity is not None else 0)
    with torch.cuda.device(dev):
        rc = _capi.lib().bsr_forward_ex(
            geom.callback, None, binning.callback, None, img.callback, None,
            P, int(degree), int(M), _ptr(t["bg"]), W, H, _ptr(t["means3D"]), _ptr(t["sh"]), _ptr(t["colors"]),
            _ptr(t["opacity"]), _ptr(t["scales"]), float(scale_modifier), _ptr(t["rotations"]), _ptr(t["cov3D"]),
            _ptr(t["view"]), _ptr(t["proj"]), _ptr(t["campos"]), float(tan_fovx), float(tan_fovy),
            int(bool(prefiltered)), out_color.data_ptr(), out_depth.data_ptr(), radii.data_ptr() if P else None,
            int(bool(debug)), _stream_handle(dev), C.byref(num_rendered), int(flags))
    _capi.check(rc, "rasterize_gaussians")
    return num_rendered.value, out_color, out_depth, radii, geom.tensor, binning.tensor, img.tensor


def _rasterize_gaussians_views_native(bg, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                      viewmatrices, projmatrices, tan_fovx, tan_fovy, image_height, image_width, sh,
                                      degree, camposes, prefiltered, debug, flags=None):
    """Forward of V views in one native call (bsr_forward_views; no reference counterpart -- the reference renders a
    camera sweep view by view): returns (num_rendered summed over the views, color [V,3,H,W], depth [V,1,H,W],
    radii [V,P]), view by view bit-identical to `_rasterize_gaussians_native`.  Forward only."""
    flags = resolve_flags() if flags is None else int(flags)
    _check_means3D(means3D)
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor; bloomscene_amd has no CPU path")
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    view = _dev_f32(viewmatrices, "viewmatrices", dev)
    proj = _dev_f32(projmatrices, "projmatrices", dev)
    campos = _dev_f32(camposes, "camposes", dev)
    if view.dim() != 3 or tuple(view.shape[1:]) != (4, 4) or proj.shape != view.shape:
        raise RuntimeError("viewmatrices / projmatrices must have dimensions (num_views, 4, 4)")
    V = view.size(0)
    if campos is None or tuple(campos.shape) != (V, 3):
        raise RuntimeError("camposes must have dimensions (num_views, 3)")
    out_color = torch.empty((V, 3, H, W), dtype=torch.float32, device=dev)
    out_depth = torch.empty((V, 1, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((V, P), dtype=torch.int32, device=dev)
    geom, binning, img = _Scratch(dev), _Scratch(dev), _Scratch(dev)
    M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
    t = dict(bg=_dev_f32(bg, "bg", dev), means3D=_dev_f32(means3D, "means3D", dev),
             colors=_dev_f32(colors, "colors_precomp", dev), opacity=_dev_f32(opacity, "opacities", dev),
             scales=_dev_f32(scales, "scales", dev), rotations=_dev_f32(rotations, "rotations", dev),
             cov3D=_dev_f32(cov3D_precomp, "cov3D_precomp", dev), sh=_dev_f32(sh, "shs", dev))
    num_rendered = C.c_int(0)
    with torch.cuda.device(dev):
        rc = _capi.lib().bsr_forward_views(
            geom.callback, None, binning.callback, None, img.callback, None,
            P, int(degree), int(M), int(V), _ptr(t["bg"]), W, H, _ptr(t["means3D"]), _ptr(t["sh"]), _ptr(t["colors"]),
            _ptr(t["opacity"]), _ptr(t["scales"]), float(scale_modifier), _ptr(t["rotations"]), _ptr(t["cov3D"]),
            _ptr(view), _ptr(proj), _ptr(campos), float(tan_fovx), float(tan_fovy),
            int(bool(prefiltered)), out_color.data_ptr(), out_depth.data_ptr(), radii.data_ptr() if P and V else None,
            int(bool(debug)), _stream_handle(dev), C.byref(num_rendered), int(flags))
    _capi.check(rc, "rasterize_gaussians_views")
    return num_rendered.value, out_color, out_depth, radii


def _rasterize_gaussians_backward_native(bg, means3D, radii, colors, scales, rotations, scale_modifier,
                                         cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                         dL_dout_depth, sh, degree, campos, geomBuffer, R, binningBuffer,
                                         imageBuffer, debug, out_depth=None, all_outputs=False, flags=None):
    """Counterpart of `_C.rasterize_gaussians_backward` = RasterizeGaussiansBackwardCUDA (RP:119-200).
    ``out_depth`` (the forward's depth image) selects the depth-gradient extension; ``flags``: the call's numerics
    (BSR_FLAG_EXACT_GRAD = the reference's per-pair operations).
    The reference also materialises dL_dconic, and dL_dcolors / dL_dcov3D even when SH / scale+rotation inputs make
    them intermediate results nobody reads (RP:154-162); here those are declined (None is returned in their place)
    unless ``all_outputs`` asks for the reference's full set."""
    flags = resolve_flags() if flags is None else int(flags)
    dev = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
    o = dict(device=dev, dtype=torch.float32)
    # every output is fully written by bsr_backward (rows of culled Gaussians = 0): no zero-fill pass
    dL_dmeans3D = torch.empty((P, 3), **o)
    dL_dmeans2D = torch.empty((P, 3), **o)
    has_colors = colors is not None and colors.numel() != 0
    has_cov = cov3D_precomp is not None and cov3D_precomp.numel() != 0
    dL_dcolors = torch.empty((P, 3), **o) if (all_outputs or has_colors) else None
    dL_dconic = torch.empty((P, 2, 2), **o) if all_outputs else None
    dL_dopacity = torch.empty((P, 1), **o)
    dL_dcov3D = torch.empty((P, 6), **o) if (all_outputs or has_cov) else None
    dL_dsh = torch.empty((P, M, 3), **o)
    has_sr = scales is not None and scales.numel() != 0
    dL_dscales = torch.empty((P, 3), **o) if has_sr else torch.zeros((P, 3), **o)
    dL_drotations = torch.empty((P, 4), **o) if has_sr else torch.zeros((P, 4), **o)
    if P != 0:
        t = dict(bg=_dev_f32(bg, "bg", dev), means3D=_dev_f32(means3D, "means3D", dev),
                 colors=_dev_f32(colors, "colors_precomp", dev), scales=_dev_f32(scales, "scales", dev),
                 rotations=_dev_f32(rotations, "rotations", dev), cov3D=_dev_f32(cov3D_precomp, "cov3D_precomp", dev),
                 view=_dev_f32(viewmatrix, "viewmatrix", dev), proj=_dev_f32(projmatrix, "projmatrix", dev),
                 sh=_dev_f32(sh, "shs", dev), campos=_dev_f32(campos, "campos", dev),
                 gcol=_dev_f32(dL_dout_color, "dL_dout_color", dev), gdep=_dev_f32(dL_dout_depth, "dL_dout_depth", dev))
        radii_c = radii.contiguous()
        head = (P, int(degree), int(M), int(R), _ptr(t["bg"]), W, H, _ptr(t["means3D"]), _ptr(t["sh"]),
                _ptr(t["colors"]), _ptr(t["scales"]), float(scale_modifier), _ptr(t["rotations"]), _ptr(t["cov3D"]),
                _ptr(t["view"]), _ptr(t["proj"]), _ptr(t["campos"]), float(tan_fovx), float(tan_fovy),
                radii_c.data_ptr(), geomBuffer.data_ptr(), binningBuffer.data_ptr() if binningBuffer.numel() else None,
                imageBuffer.data_ptr())
        tail = (_ptr(t["gcol"]), _ptr(t["gdep"]), dL_dmeans2D.data_ptr(),
                _ptr(dL_dconic), dL_dopacity.data_ptr(), _ptr(dL_dcolors), dL_dmeans3D.data_ptr(),
                _ptr(dL_dcov3D), dL_dsh.data_ptr() if M else None, dL_dscales.data_ptr(),
                dL_drotations.data_ptr(), int(bool(debug)), _stream_handle(dev))
        od = None if out_depth is None else _dev_f32(out_depth, "out_depth", dev)
        with torch.cuda.device(dev):
            rc = _capi.lib().bsr_backward_ex(*head, _ptr(od), *tail, int(flags))
        _capi.check(rc, "rasterize_gaussians_backward")
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def _mark_visible_native(means3D, viewmatrix, projmatrix):
    """Counterpart of `_C.mark_visible` = markVisible (RP:202-221)."""
    dev = means3D.device
    if not means3D.is_cuda:
        raise RuntimeError("positions must be a GPU tensor; bloomscene_amd has no CPU path")
    P = means3D.size(0)
    present = torch.full((P,), False, dtype=torch.bool, device=dev)
    if P != 0:
        m, v, p = _dev_f32(means3D, "means3D", dev), _dev_f32(viewmatrix, "viewmatrix", dev), _dev_f32(projmatrix, "projmatrix", dev)
        with torch.cuda.device(dev):
            rc = _capi.lib().bsr_mark_visible(P, m.data_ptr(), v.data_ptr(), p.data_ptr(), present.data_ptr(),
                                              _stream_handle(dev))
        _capi.check(rc, "mark_visible")
    return present


def _rasterize_gaussians_filter_native(means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                       projmatrix, tan_fovx, tan_fovy, image_height, image_width, prefiltered, debug):
    """Counterpart of `_C.rasterize_aussians_filter` [sic] = RasterizeGaussiansfilterCUDA (RP:224-288)."""
    _check_means3D(means3D)
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor; bloomscene_amd has no CPU path")
    dev = means3D.device
    P = means3D.size(0)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    if P != 0:
        m = _dev_f32(means3D, "means3D", dev)
        s, r = _dev_f32(scales, "scales", dev), _dev_f32(rotations, "rotations", dev)
        c = _dev_f32(cov3D_precomp, "cov3D_precomp", dev)
        v, p = _dev_f32(viewmatrix, "viewmatrix", dev), _dev_f32(projmatrix, "projmatrix", dev)
        with torch.cuda.device(dev):
            rc = _capi.lib().bsr_visible_filter(
                P, 0, int(image_width), int(image_height), m.data_ptr(), _ptr(s), float(scale_modifier), _ptr(r),
                _ptr(c), v.data_ptr(), p.data_ptr(), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                radii.data_ptr(), int(bool(debug)), _stream_handle(dev))
        _capi.check(rc, "rasterize_gaussians_filter")
    return radii


def _rasterize_gaussians_filter_indices_native(means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                               projmatrix, tan_fovx, tan_fovy, image_height, image_width, prefiltered,
                                               debug):
    """visible_filter plus the ascending index list of the visible points (bsr_visible_filter_indices):
    -> (radii int32 [P], visible_idx int64 [n_visible])."""
    _check_means3D(means3D)
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor; bloomscene_amd has no CPU path")
    dev = means3D.device
    P = means3D.size(0)
    radii = torch.empty((P,), dtype=torch.int32, device=dev)
    idx = torch.empty((P,), dtype=torch.int32, device=dev)
    n = C.c_int(0)
    if P != 0:
        lib = _capi.lib()
        m = _dev_f32(means3D, "means3D", dev)
        s, r = _dev_f32(scales, "scales", dev), _dev_f32(rotations, "rotations", dev)
        c = _dev_f32(cov3D_precomp, "cov3D_precomp", dev)
        v, p = _dev_f32(viewmatrix, "viewmatrix", dev), _dev_f32(projmatrix, "projmatrix", dev)
        scratch = torch.empty(lib.bsr_visible_scratch_bytes(P), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = lib.bsr_visible_filter_indices(
                P, 0, int(image_width), int(image_height), m.data_ptr(), _ptr(s), float(scale_modifier), _ptr(r),
                _ptr(c), v.data_ptr(), p.data_ptr(), float(tan_fovx), float(tan_fovy), int(bool(prefiltered)),
                radii.data_ptr(), idx.data_ptr(), scratch.data_ptr(), C.byref(n), int(bool(debug)), _stream_handle(dev))
        _capi.check(rc, "rasterize_gaussians_filter_indices")
    return radii, idx[:n.value].long()


def _rasterize_gaussians_filter_views_native(means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrices,
                                             projmatrices, tan_fovx, tan_fovy, image_height, image_width, debug):
    """visible_filter for V cameras in one pass (bsr_visible_filter_views): -> int32 [V, P]."""
    _check_means3D(means3D)
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor; bloomscene_amd has no CPU path")
    dev = means3D.device
    P = means3D.size(0)
    if viewmatrices.dim() != 3 or tuple(viewmatrices.shape[1:]) != (4, 4) or viewmatrices.shape != projmatrices.shape:
        raise RuntimeError("viewmatrices and projmatrices must both have dimensions (num_views, 4, 4)")
    V = viewmatrices.size(0)
    radii = torch.empty((V, P), dtype=torch.int32, device=dev)
    if P != 0 and V != 0:
        m = _dev_f32(means3D, "means3D", dev)
        s, r = _dev_f32(scales, "scales", dev), _dev_f32(rotations, "rotations", dev)
        c = _dev_f32(cov3D_precomp, "cov3D_precomp", dev)
        v, p = _dev_f32(viewmatrices, "viewmatrices", dev), _dev_f32(projmatrices, "projmatrices", dev)
        with torch.cuda.device(dev):
            rc = _capi.lib().bsr_visible_filter_views(
                P, V, int(image_width), int(image_height), m.data_ptr(), _ptr(s), float(scale_modifier), _ptr(r),
                _ptr(c), v.data_ptr(), p.data_ptr(), float(tan_fovx), float(tan_fovy), radii.data_ptr(),
                int(bool(debug)), _stream_handle(dev))
        _capi.check(rc, "rasterize_gaussians_filter_views")
    return radii


def _rasterize_gaussians_filter_groups_native(means3D, scales, rotations, scale_modifier, cov3D_precomp, viewmatrices,
                                              projmatrices, tan_fovx, tan_fovy, image_height, image_width,
                                              group_of_view, n_groups, debug, return_counts=False):
    """Per-group visibility (bsr_visible_filter_groups): -> bool [n_groups, P], row g = "some view of group g has
    radii > 0"; ``group_of_view`` int32 [V] on the device.  ``return_counts``: also the ones per row, int32 [n_groups]
    on the device (accumulated by the kernel)."""
    _check_means3D(means3D)
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a GPU tensor; bloomscene_amd has no CPU path")
    dev = means3D.device
    P = means3D.size(0)
    if viewmatrices.dim() != 3 or tuple(viewmatrices.shape[1:]) != (4, 4) or viewmatrices.shape != projmatrices.shape:
        raise RuntimeError("viewmatrices and projmatrices must both have dimensions (num_views, 4, 4)")
    V = viewmatrices.size(0)
    if group_of_view.numel() != V:
        raise RuntimeError("group_of_view must hold one group id per view")
    mask = torch.empty((int(n_groups), P), dtype=torch.bool, device=dev)
    # (every count is written by the call; only the P = 0 shortcut below leaves the library out)
    counts = (torch.empty if P != 0 else torch.zeros)((int(n_groups),), dtype=torch.int32, device=dev) if return_counts else None
    if P != 0 and n_groups != 0:
        m = _dev_f32(means3D, "means3D", dev)
        s, r = _dev_f32(scales, "scales", dev), _dev_f32(rotations, "rotations", dev)
        c = _dev_f32(cov3D_precomp, "cov3D_precomp", dev)
        v, p = _dev_f32(viewmatrices, "viewmatrices", dev), _dev_f32(projmatrices, "projmatrices", dev)
        gv = group_of_view.to(device=dev, dtype=torch.int32).contiguous()
        lib = _capi.lib()
        # the kernel's per-workgroup partial counts: the caller's scratch, like every other byte the library uses
        scratch = torch.empty(lib.bsr_visible_groups_scratch_bytes(P, int(n_groups)), dtype=torch.uint8,
                              device=dev) if return_counts else None
        with torch.cuda.device(dev):
            rc = lib.bsr_visible_filter_groups(
                P, V, int(n_groups), int(image_width), int(image_height), m.data_ptr(), _ptr(s), float(scale_modifier),
                _ptr(r), _ptr(c), _ptr(v), _ptr(p), float(tan_fovx), float(tan_fovy), gv.data_ptr() if V else None,
                mask.data_ptr(), counts.data_ptr() if return_counts else None, _ptr(scratch), int(bool(debug)),
                _stream_handle(dev))
        _capi.check(rc, "rasterize_gaussians_filter_groups")
    return (mask, counts) if return_counts else mask


def _gather_rows_native(tensors, idx, idx_stride=1, rows=None, packed=True, debug=False):
    """Rows ``idx`` (int64 on the device, every ``idx_stride``-th element) of up to eight fp32 tensors [P, ...] in one
    pass.  ``packed``: one [R, sum of row widths] matrix, the tensors' rows side by side (bsr_pack_rows); otherwise a
    list of tensors [R, ...] with the sources' trailing shapes (bsr_gather_rows)."""
    import ctypes as C
    if not 1 <= len(tensors) <= 8:
        raise RuntimeError("gather_rows takes 1..8 tensors")
    dev = tensors[0].device
    if dev.type != "cuda":
        raise RuntimeError("gather_rows needs GPU tensors; bloomscene_amd has no CPU path")
    if idx.dtype != torch.int64 or idx.device != dev or not idx.is_contiguous():
        raise RuntimeError("idx must be a contiguous int64 tensor on the tensors' device")
    idx_stride = int(idx_stride)
    R = int(rows) if rows is not None else (idx.numel() + idx_stride - 1) // idx_stride
    if R < 0 or idx_stride < 1 or (R > 0 and (R - 1) * idx_stride + 1 > idx.numel()):
        raise RuntimeError("idx holds fewer than rows elements at this stride")
    widths = [int(torch.Size(t.shape[1:]).numel()) for t in tensors]
    if any(t.shape[0] != tensors[0].shape[0] for t in tensors):
        raise RuntimeError("gather_rows: the tensors must have the same number of rows")
    srcs = [_dev_f32(t.detach(), "gather_rows source", dev) for t in tensors] if R else []
    if R and any(t is None for t in srcs):
        raise RuntimeError("gather_rows: rows requested from an empty tensor")
    if packed:
        out = torch.empty((R, sum(widths)), dtype=torch.float32, device=dev)
    else:
        out = [torch.empty((R,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev) for t in tensors]
    if R:
        n = len(srcs)
        table = (C.c_void_p * n)(*[t.data_ptr() for t in srcs])
        wtab = (C.c_int * n)(*widths)
        with torch.cuda.device(dev):
            if packed:
                rc = _capi.lib().bsr_pack_rows(R, int(tensors[0].shape[0]), n, table, wtab, idx.data_ptr(), idx_stride,
                                               out.data_ptr(), int(bool(debug)), _stream_handle(dev))
            else:
                rc = _capi.lib().bsr_gather_rows(R, int(tensors[0].shape[0]), n, table, wtab, idx.data_ptr(), idx_stride,
                                                 (C.c_void_p * n)(*[t.data_ptr() for t in out]), int(bool(debug)),
                                                 _stream_handle(dev))
        _capi.check(rc, "gather_rows")
    return out


def _pack_rows_native(tensors, idx, idx_stride=1, rows=None, debug=False):
    """bsr_pack_rows: [R, sum of row widths] fp32 (see _gather_rows_native)."""
    return _gather_rows_native(tensors, idx, idx_stride, rows, True, debug)


def read_counts(image_buffer, image_height, image_width):
    """(kept, num_rendered) of the forward call that filled ``image_buffer`` (the third scratch tensor the native forward
    returns / an autograd node saves): bsr_read_counts, blocks on the current stream.  For capacity-mode callers."""
    kept, R = C.c_int(0), C.c_int(0)
    dev = image_buffer.device
    with torch.cuda.device(dev):
        rc = _capi.lib().bsr_read_counts(image_buffer.data_ptr(), int(image_width), int(image_height), _stream_handle(dev),
                                         C.byref(kept), C.byref(R))
    _capi.check(rc, "bsr_read_counts")
    return kept.value, R.value


def check_deferred():
    """Raises if the calling thread's last capacity-mode forward overflowed its capacity (bsr_check_deferred)."""
    _capi.check(_capi.lib().bsr_check_deferred(), "deferred check of the previous no-readback forward")


# ------------------------------------------------------------------ autograd (PYW:21-156)
def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings, depth_gradient=False, return_final_T=False, flags=None, capacity=None):
    """``flags``: BSR_FLAG_* word of this call; None = the calling thread's `numerics(...)` context (default 0).
    ``capacity``: see GaussianRasterizer."""
    out = _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                    cov3Ds_precomp, raster_settings, depth_gradient, return_final_T,
                                    resolve_flags() if flags is None else int(flags), capacity)
    return out if return_final_T else out[:3]


class _RasterizeGaussians(torch.autograd.Function):

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, depth_gradient=False, return_final_T=False, flags=0, capacity=None):
        # same argument order as the reference hands to its C++ lib (PYW:60-80)
        args = (
            raster_settings.bg, means3D, colors_precomp, opacities, scales, rotations,
            raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix,
            raster_settings.tanfovx, raster_settings.tanfovy, raster_settings.image_height,
            raster_settings.image_width, sh, raster_settings.sh_degree, raster_settings.campos,
            raster_settings.prefiltered, raster_settings.debug,
        )
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # copy them before they can be corrupted (PYW:84)
            try:
                num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer = \
                    _rasterize_gaussians_native(*args, flags=flags, capacity=capacity)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, color, depth, radii, geomBuffer, binningBuffer, imgBuffer = \
                _rasterize_gaussians_native(*args, flags=flags, capacity=capacity)

        # accumulated opacity of the call (extension, GaussianRasterizer.forward(return_alpha=True)): final_T lives in the
        # image buffer, where the library says it is; the view is handed out as a fourth, non-differentiable output
        if return_final_T:
            n_pix = raster_settings.image_height * raster_settings.image_width
            if imgBuffer.numel() >= 4 * n_pix:
                off = int(_capi.lib().bsr_transmittance_offset(imgBuffer.data_ptr()))
                final_T = imgBuffer[off:off + 4 * n_pix].view(torch.float32).view(
                    1, raster_settings.image_height, raster_settings.image_width)
            else:   # P == 0: nothing was rendered, nothing absorbed
                final_T = torch.ones((1, raster_settings.image_height, raster_settings.image_width),
                                     dtype=torch.float32, device=color.device)
        # radii is an index output, and unused outputs need no zero gradients: without these two lines autograd
        # fills an int32 [P] zero tensor for grad_radii on every backward (the reference pays that fill, PYW:101)
        if return_final_T:
            ctx.mark_non_differentiable(radii, final_T)
        else:
            ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.depth_gradient = bool(depth_gradient)
        ctx.flags = int(flags)   # the backward runs on an autograd worker thread: the mode travels with the node
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        if ctx.depth_gradient:
            ctx.out_depth = depth.detach().clone()   # the extension differentiates through this image
        return color, radii, depth, (final_T if return_final_T else None)

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_depth, _grad_final_T=None):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer = \
            ctx.saved_tensors
        if grad_out_color is None:  # only the depth output was used downstream; it carries no gradient
            grad_out_color = torch.zeros((3, raster_settings.image_height, raster_settings.image_width),
                                         dtype=torch.float32, device=means3D.device)
        if grad_depth is None:
            grad_depth = torch.zeros((1, raster_settings.image_height, raster_settings.image_width),
                                     dtype=torch.float32, device=means3D.device)
        args = (raster_settings.bg, means3D, radii, colors_precomp, scales, rotations,
                raster_settings.scale_modifier, cov3Ds_precomp, raster_settings.viewmatrix,
                raster_settings.projmatrix, raster_settings.tanfovx, raster_settings.tanfovy, grad_out_color,
                grad_depth, sh, raster_settings.sh_degree, raster_settings.campos, geomBuffer, num_rendered,
                binningBuffer, imgBuffer, raster_settings.debug)
        if ctx.depth_gradient:
            args = args + (ctx.out_depth,)
        kw = dict(flags=ctx.flags)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, \
                    grad_scales, grad_rotations = _rasterize_gaussians_backward_native(*args, **kw)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, \
                grad_scales, grad_rotations = _rasterize_gaussians_backward_native(*args, **kw)

        # Absent optional inputs were empty tensors: their gradient must have the input's (empty)
        # shape for autograd's shape check; the reference relies on older, laxer torch here.
        def _fit(g, inp):
            return g if (inp.numel() != 0 and g is not None) else None

        grads = (
            grad_means3D,
            grad_means2D,
            _fit(grad_sh, sh),
            _fit(grad_colors_precomp, colors_precomp),
            grad_opacities,
            _fit(grad_scales, scales),
            _fit(grad_rotations, rotations),
            _fit(grad_cov3Ds_precomp, cov3Ds_precomp),
            None,
            None,
            None,
            None,
            None,
        )
        return grads


class GaussianRasterizer(nn.Module):  # PYW:172-249
    def __init__(self, raster_settings, depth_gradient=False, exact_exp=None, strict_gradients=None, capacity=None):
        """``depth_gradient`` (extension, default off = the reference's behaviour): also backpropagate the
        gradient of the depth output.  The reference accepts grad_depth and drops it (backward.cu:457-463,
        539-554), so depth losses never move the Gaussians there; see include/bloomscene_rast.h
        bsr_backward_depth.

        Numerics of THIS rasterizer's calls (per call through the C ABI's flags, include/bloomscene_rast.h BSR_FLAG_*;
        None = whatever the calling thread's ``bloomscene_amd.numerics(...)`` context says, by default the fast path):
        ``exact_exp=True`` -- the pinned exp on every evaluation of the forward blend (bit-equal to the CPU oracle);
        ``strict_gradients=True`` -- the backward tile walk performs the reference's per-pair operations
        (backward.cu:521,527-536,557,561-583), which meets SURVEY.md 8(d)'s elementwise gradient bar at ~1.8x the
        cost of that kernel.

        ``capacity`` (extension, default None = the reference's behaviour): a number of tile instances.  The reference
        blocks the host in every forward on the read-back of num_rendered (rasterizer_impl.cu:282) and so does the
        default path; with a capacity the forward never waits for the GPU (BSR_FLAG_NO_READBACK): the binning scratch is
        sized for ``capacity``, a frame that needs more comes back as NaN and the NEXT forward of the thread raises
        (``read_counts(...)`` tells how much was needed).  A warmed-up forward + backward in this mode can be captured
        into a HIP graph (torch.cuda.graph)."""
        super().__init__()
        self.capacity = None if capacity is None else int(capacity)
        self.raster_settings = raster_settings
        self.depth_gradient = bool(depth_gradient)
        self.exact_exp = exact_exp
        self.strict_gradients = strict_gradients

    def _flags(self):
        return resolve_flags(self.exact_exp, self.strict_gradients)

    def markVisible(self, positions):
        # Mark visible points (based on frustum culling for camera) with a boolean
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _mark_visible_native(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, return_alpha=False):
        """Same call as the reference (PYW:188-221) -> (color, radii, depth).  Extension, off by default:
        ``return_alpha=True`` appends ``alpha = 1 - final_T`` [1,H,W] (accumulated opacity; the reference
        keeps final_T only inside its opaque image buffer, forward.cu:459).  It carries no gradient."""
        raster_settings = self.raster_settings

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])

        if not return_alpha:
            return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                       cov3D_precomp, raster_settings, self.depth_gradient, flags=self._flags(),
                                       capacity=self.capacity)
        color, radii, depth, final_T = rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales,
                                                           rotations, cov3D_precomp, raster_settings,
                                                           self.depth_gradient, return_final_T=True,
                                                           flags=self._flags(), capacity=self.capacity)
        return color, radii, depth, (1.0 - final_T).detach()

    def visible_filter_indices(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        """EXTENSION: ``visible_filter`` that also returns the ascending indices of the visible points,
        ``(radii int32 [P], idx int64 [n_visible])`` with ``idx == (radii > 0).nonzero().squeeze(1)``; one native call
        and one host synchronisation instead of a nonzero() pass per boolean index (GR:33-43)."""
        rs = self.raster_settings
        e = torch.Tensor([])
        with torch.no_grad():
            return _rasterize_gaussians_filter_indices_native(
                means3D, e if scales is None else scales, e if rotations is None else rotations, rs.scale_modifier,
                e if cov3D_precomp is None else cov3D_precomp, rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy,
                rs.image_height, rs.image_width, rs.prefiltered, rs.debug)

    def visible_filter(self, means3D, scales=None, rotations=None, cov3D_precomp=None):
        raster_settings = self.raster_settings
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        with torch.no_grad():
            radii = _rasterize_gaussians_filter_native(
                means3D, scales, rotations, raster_settings.scale_modifier, cov3D_precomp,
                raster_settings.viewmatrix, raster_settings.projmatrix, raster_settings.tanfovx,
                raster_settings.tanfovy, raster_settings.image_height, raster_settings.image_width,
                raster_settings.prefiltered, raster_settings.debug)
        return radii
