"""ctypes/numpy front-end of the CPU oracle (``oracle/bsr_oracle.c``).

TEST INFRASTRUCTURE ONLY -- see the header of ``bsr_oracle.c``.  Imported by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``; never by
``bloomscene_amd``.  PARITY UNPINNED (the reference has no tests/goldens and cannot be built
or imported here).

The orchestration below restates ``CudaRasterizer::Rasterizer::forward/backward/
visible_filter/markVisible`` (cuda_rasterizer/rasterizer_impl.cu:141-504) and the torch glue
``RasterizeGaussiansCUDA`` / ``RasterizeGaussiansBackwardCUDA`` (rasterize_points.cu:35-200) of
``/root/reference/submodules/depth-diff-gaussian-rasterization``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BSR_ORACLE_LIB: load another build of the oracle instead (tests/test_oracle_mutations.py points it at deliberately
# broken copies to prove that the checks would catch a transcription slip)
_LIB_PATH = os.environ.get("BSR_ORACLE_LIB") or os.path.join(_HERE, "libbsr_oracle.so")
_lib = None

BLOCK_X = 16
BLOCK_Y = 16


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "bsr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libbsr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.environ.get("BSR_ORACLE_LIB"):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.bsro_expf.restype = C.c_float
        _lib.bsro_expf.argtypes = [C.c_float]
        _lib.bsro_inclusive_sum.restype = C.c_uint32
        _lib.bsro_get_higher_msb.restype = C.c_uint32
        _lib.bsro_get_higher_msb.argtypes = [C.c_uint32]
        _lib.bsro_preprocess.restype = C.c_int
    return _lib


def _f32(a, shape=None):
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a):
    """Pointer or NULL (absent optional input == nullptr, python wrapper :198-208)."""
    if a is None or a.size == 0:
        return None
    return a.ctypes.data_as(C.c_void_p)


def expf(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty_like(x)
    lib().bsro_expf_array(C.c_int(x.size), _p(x), y.ctypes.data_as(C.c_void_p))
    return y


def make_settings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix,
                  sh_degree, campos, prefiltered=False, debug=False):
    """Same 12 fields as GaussianRasterizationSettings (python wrapper :158-170), numpy-valued."""
    return SimpleNamespace(
        image_height=int(image_height), image_width=int(image_width), tanfovx=float(tanfovx),
        tanfovy=float(tanfovy), bg=_f32(bg, (3,)), scale_modifier=float(scale_modifier),
        viewmatrix=_f32(viewmatrix, (16,)), projmatrix=_f32(projmatrix, (16,)), sh_degree=int(sh_degree),
        campos=_f32(campos, (3,)), prefiltered=bool(prefiltered), debug=bool(debug))


def settings_from(rs):
    """Accept a GaussianRasterizationSettings-like object holding torch tensors."""
    return make_settings(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.bg, rs.scale_modifier,
                         rs.viewmatrix, rs.projmatrix, rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)


def mark_visible(positions, rs):
    """Rasterizer::markVisible (rasterizer_impl.cu:141-153)."""
    pos = _f32(positions).reshape(-1, 3)
    P = pos.shape[0]
    present = np.zeros(P, dtype=np.uint8)
    if P:
        lib().bsro_mark_visible(C.c_int(P), _p(pos), _p(rs.viewmatrix), _p(rs.projmatrix), _p(present))
    return present.astype(bool)


def _preprocess(rs, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, filter_only):
    P = means3D.shape[0]
    W, H = rs.image_width, rs.image_height
    M = 0 if shs is None or shs.size == 0 else shs.shape[1]
    st = SimpleNamespace(P=P, M=M, W=W, H=H)
    st.radii = np.zeros(P, dtype=np.int32)
    st.means2D = np.zeros((P, 2), dtype=np.float32)
    st.depths = np.zeros(P, dtype=np.float32)
    st.cov3D = np.zeros((P, 6), dtype=np.float32)
    st.rgb = np.zeros((P, 3), dtype=np.float32)
    st.conic_opacity = np.zeros((P, 4), dtype=np.float32)
    st.clamped = np.zeros((P, 3), dtype=np.uint8)
    st.tiles_touched = np.zeros(P, dtype=np.uint32)
    if P == 0:
        return st
    rc = lib().bsro_preprocess(
        C.c_int(P), C.c_int(rs.sh_degree), C.c_int(M), _p(means3D), _p(scales), C.c_float(rs.scale_modifier),
        _p(rotations), _p(opacities), _p(shs), _p(st.clamped), _p(cov3D_precomp), _p(colors_precomp),
        _p(rs.viewmatrix), _p(rs.projmatrix), _p(rs.campos), C.c_int(W), C.c_int(H), C.c_float(rs.tanfovx),
        C.c_float(rs.tanfovy), _p(st.radii), _p(st.means2D), _p(st.depths), _p(st.cov3D), _p(st.rgb),
        _p(st.conic_opacity), _p(st.tiles_touched), C.c_int(int(rs.prefiltered)), C.c_int(int(filter_only)))
    if rc != 0:
        raise RuntimeError("Point is filtered although prefiltered is set. This shouldn't happen!")
    return st


def _check_inputs(means3D):
    if means3D.ndim != 2 or means3D.shape[1] != 3:
        raise ValueError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:57-59


def visible_filter(rs, means3D, scales=None, rotations=None, cov3D_precomp=None):
    """Rasterizer::visible_filter (rasterizer_impl.cu:342-398) -> radii int32[P]."""
    means3D = _f32(means3D)
    _check_inputs(means3D)
    st = _preprocess(rs, means3D, None, None, None, _f32(scales), _f32(rotations), _f32(cov3D_precomp), True)
    return st.radii


def forward(rs, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
            cov3D_precomp=None):
    """Rasterizer::forward (rasterizer_impl.cu:198-339).  Returns a namespace with the three
    public outputs (color [3,H,W], radii [P], depth [1,H,W]) and every intermediate."""
    means3D = _f32(means3D)
    _check_inputs(means3D)
    opacities = _f32(opacities)
    shs, colors_precomp = _f32(shs), _f32(colors_precomp)
    scales, rotations, cov3D_precomp = _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    L = lib()
    st = _preprocess(rs, means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, False)
    st.inputs = SimpleNamespace(means3D=means3D, opacities=opacities, shs=shs, colors_precomp=colors_precomp,
                                scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    st.rs = rs
    P, W, H = st.P, st.W, st.H
    gx, gy = (W + BLOCK_X - 1) // BLOCK_X, (H + BLOCK_Y - 1) // BLOCK_Y
    st.grid = (gx, gy)
    st.color = np.zeros((3, H, W), dtype=np.float32)
    st.depth = np.zeros((1, H, W), dtype=np.float32)
    st.final_T = np.zeros(H * W, dtype=np.float32)
    st.n_contrib = np.zeros(H * W, dtype=np.uint32)
    st.ranges = np.zeros((gx * gy, 2), dtype=np.uint32)
    st.point_offsets = np.zeros(P, dtype=np.uint32)
    st.num_rendered = 0
    st.point_list = np.zeros(0, dtype=np.uint32)
    st.point_list_keys = np.zeros(0, dtype=np.uint64)
    if P == 0:  # rasterize_points.cu:82
        return st
    R = int(L.bsro_inclusive_sum(C.c_int(P), _p(st.tiles_touched), _p(st.point_offsets)))
    st.num_rendered = R
    keys_u = np.zeros(max(R, 1), dtype=np.uint64)
    vals_u = np.zeros(max(R, 1), dtype=np.uint32)
    L.bsro_duplicate_with_keys(C.c_int(P), _p(st.means2D), _p(st.depths), _p(st.point_offsets), _p(keys_u),
                               _p(vals_u), _p(st.radii), C.c_int(gx), C.c_int(gy))
    bit = int(L.bsro_get_higher_msb(C.c_uint32(gx * gy)))
    keys = np.zeros(max(R, 1), dtype=np.uint64)
    vals = np.zeros(max(R, 1), dtype=np.uint32)
    L.bsro_sort_pairs(C.c_int(R), _p(keys_u), _p(keys), _p(vals_u), _p(vals), C.c_int(32 + bit))
    if R > 0:
        L.bsro_identify_tile_ranges(C.c_int(R), _p(keys), _p(st.ranges))
    st.point_list = vals[:R]
    st.point_list_keys = keys[:R]
    feat = colors_precomp if colors_precomp is not None else st.rgb
    st.features = feat
    L.bsro_render_forward(_p(st.ranges), _p(vals), C.c_int(W), C.c_int(H), _p(st.means2D), _p(feat),
                          _p(st.depths), _p(st.conic_opacity), _p(st.final_T), _p(st.n_contrib), _p(rs.bg),
                          _p(st.color), _p(st.depth))
    return st


def backward(st, grad_color, grad_depth=None, want_abs_sums=False, depth_gradient=False, f32_sums=False, segment=0):
    """Rasterizer::backward (rasterizer_impl.cu:403-504) + RasterizeGaussiansBackwardCUDA
    (rasterize_points.cu:119-200).  ``grad_depth`` is accepted and ignored, like the reference --
    unless ``depth_gradient=True``, the opt-in EXTENSION that adds the true
    derivative of the normalised depth target (bsro_render_backward_depth; pinned by autograd, not
    by the reference).  Returns the 8 gradients in the order of the reference tuple plus the
    internal dL_dconic."""
    L = lib()
    rs, inp = st.rs, st.inputs
    P, M, W, H = st.P, st.M, st.W, st.H
    g = SimpleNamespace()
    g.dL_dmeans3D = np.zeros((P, 3), dtype=np.float32)
    g.dL_dmeans2D = np.zeros((P, 3), dtype=np.float32)
    g.dL_dcolors = np.zeros((P, 3), dtype=np.float32)
    g.dL_dconic = np.zeros((P, 2, 2), dtype=np.float32)
    g.dL_dopacity = np.zeros((P, 1), dtype=np.float32)
    g.dL_dcov3D = np.zeros((P, 6), dtype=np.float32)
    g.dL_dsh = np.zeros((P, M, 3), dtype=np.float32)
    g.dL_dscales = np.zeros((P, 3), dtype=np.float32)
    g.dL_drotations = np.zeros((P, 4), dtype=np.float32)
    g.abs_sums = np.zeros((P, 9), dtype=np.float32) if want_abs_sums else None
    if P == 0:
        return g
    grad_color = _f32(grad_color, (3, H, W))
    grad_depth = _f32(grad_depth) if grad_depth is not None else np.zeros((1, H, W), dtype=np.float32)
    focal_y = np.float32(H) / (np.float32(2.0) * np.float32(rs.tanfovy))
    focal_x = np.float32(W) / (np.float32(2.0) * np.float32(rs.tanfovx))
    R = st.num_rendered
    plist = st.point_list if R > 0 else np.zeros(1, dtype=np.uint32)
    # f32_sums: the pair sums added in binary32 in one fixed order (one legal outcome of the reference's float
    # atomicAdds) instead of the order-free binary64 definition; only to measure the spread between legal outcomes
    L.bsro_set_sum_mode(C.c_int(1 if f32_sums else 0))
    # segment > 0: measurement only -- the recurrences restarted from forward checkpoints every `segment` list entries
    # (bsr_oracle.c: bsro_set_backward_segment; docs/EXPERIMENTS.md round 6)
    L.bsro_set_backward_segment(C.c_int(int(segment)))
    try:
        L.bsro_render_backward(
            C.c_int(P), C.c_int(R), _p(st.ranges), _p(plist), C.c_int(W), C.c_int(H), _p(rs.bg), _p(st.means2D),
            _p(st.conic_opacity), _p(st.features), _p(st.final_T), _p(st.n_contrib), _p(grad_color), _p(grad_depth),
            _p(g.dL_dmeans2D), _p(g.dL_dconic), _p(g.dL_dopacity), _p(g.dL_dcolors),
            _p(g.abs_sums) if want_abs_sums else None)
    finally:
        L.bsro_set_sum_mode(C.c_int(0))
        L.bsro_set_backward_segment(C.c_int(0))
    if not depth_gradient:
        return backward_chain(st, g)
    g.dL_dz = np.zeros((P,), dtype=np.float32)
    L.bsro_render_backward_depth(
        C.c_int(P), C.c_int(R), _p(st.ranges), _p(plist), C.c_int(W), C.c_int(H), _p(st.means2D),
        _p(st.conic_opacity), _p(st.depths), _p(st.final_T), _p(st.n_contrib), _p(st.depth), _p(grad_depth),
        _p(g.dL_dmeans2D), _p(g.dL_dconic), _p(g.dL_dopacity), _p(g.dL_dz))
    backward_chain(st, g)
    # view z = vm[2] x + vm[6] y + vm[10] z + vm[14] (flat, column-vector convention of auxiliary.h:58-66)
    vm = np.asarray(rs.viewmatrix, dtype=np.float32).reshape(-1)
    g.dL_dmeans3D += g.dL_dz[:, None] * np.array([vm[2], vm[6], vm[10]], dtype=np.float32)[None, :]
    return g


def backward_chain(st, g):
    """The per-Gaussian part of the backward (computeCov2DCUDA + preprocessCUDA,
    backward.cu:144-274,346-396) applied to the accumulators already present in ``g``
    (dL_dmeans2D, dL_dconic, dL_dopacity, dL_dcolors).  Split out so tests can feed it the HIP
    path's accumulators and check the chain in isolation from the unordered float sums."""
    L = lib()
    rs, inp = st.rs, st.inputs
    P, M, W, H = st.P, st.M, st.W, st.H
    focal_y = np.float32(H) / (np.float32(2.0) * np.float32(rs.tanfovy))
    focal_x = np.float32(W) / (np.float32(2.0) * np.float32(rs.tanfovx))
    cov3D = inp.cov3D_precomp if inp.cov3D_precomp is not None else st.cov3D
    L.bsro_backward_cov2d(C.c_int(P), _p(inp.means3D), _p(st.radii), _p(cov3D), C.c_float(focal_x),
                          C.c_float(focal_y), C.c_float(rs.tanfovx), C.c_float(rs.tanfovy), _p(rs.viewmatrix),
                          _p(g.dL_dconic), _p(g.dL_dmeans3D), _p(g.dL_dcov3D))
    L.bsro_backward_preprocess(
        C.c_int(P), C.c_int(rs.sh_degree), C.c_int(M), _p(inp.means3D), _p(st.radii), _p(inp.shs), _p(st.clamped),
        _p(inp.scales), _p(inp.rotations), C.c_float(rs.scale_modifier), _p(rs.projmatrix), _p(rs.campos),
        _p(g.dL_dmeans2D), _p(g.dL_dmeans3D), _p(g.dL_dcolors), _p(g.dL_dcov3D), _p(g.dL_dsh), _p(g.dL_dscales),
        _p(g.dL_drotations))
    return g


def empty_grads(st):
    """Zero-filled gradient namespace (RasterizeGaussiansBackwardCUDA, rasterize_points.cu:154-162)."""
    P, M = st.P, st.M
    g = SimpleNamespace()
    g.dL_dmeans3D = np.zeros((P, 3), dtype=np.float32)
    g.dL_dmeans2D = np.zeros((P, 3), dtype=np.float32)
    g.dL_dcolors = np.zeros((P, 3), dtype=np.float32)
    g.dL_dconic = np.zeros((P, 2, 2), dtype=np.float32)
    g.dL_dopacity = np.zeros((P, 1), dtype=np.float32)
    g.dL_dcov3D = np.zeros((P, 6), dtype=np.float32)
    g.dL_dsh = np.zeros((P, M, 3), dtype=np.float32)
    g.dL_dscales = np.zeros((P, 3), dtype=np.float32)
    g.dL_drotations = np.zeros((P, 4), dtype=np.float32)
    g.abs_sums = None
    return g
