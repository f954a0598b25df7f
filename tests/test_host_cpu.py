"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the
python mirror has the reference's API surface and error behaviour, camera conventions, view
sharding, and the sort network used by the per-tile sort.  No compute call reaches the GPU."""
import math
import os
import re

import numpy as np
import pytest
import torch

import helpers as Hh
from bloomscene_amd import GaussianRasterizationSettings, GaussianRasterizer, _capi, cameras, views

ROOT = Hh.ROOT


def test_library_loads_and_exports_every_declared_symbol():
    hdr = "".join(open(os.path.join(ROOT, "include", h)).read() for h in sorted(os.listdir(os.path.join(ROOT, "include"))))
    hdr_nocomment = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(bsr_[a-z_]+)\s*\(", hdr_nocomment))
    declared -= {"bsr_alloc_fn"}
    assert {"bsr_forward", "bsr_backward", "bsr_visible_filter", "bsr_mark_visible", "bsr_last_error",
            "bsr_version", "bsr_anchor_select", "bsr_anchor_expand", "bsr_anchor_expand_backward"} <= declared
    lib = _capi.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/bloomscene_rast.h but not exported"
    assert declared == set(_capi.SIGNATURES), "ctypes table and header disagree"
    assert lib.bsr_version() == 4
    assert lib.bsr_last_error() == b""
    # scratch sizing (reference required<T>(n), rasterizer_impl.h:68-73): monotone, 256-B granular
    assert lib.bsr_geometry_bytes(0) < lib.bsr_geometry_bytes(1000) < lib.bsr_geometry_bytes(2000)
    # record 64, instance offset 4, kept mask 8, rect 8, clamp bits 1, depth 4 per Gaussian (cov3D is not kept: round 5)
    assert lib.bsr_geometry_bytes(1000) >= 1000 * (64 + 4 + 8 + 8 + 1 + 4) + 256 * 8 * 4
    assert lib.bsr_binning_bytes(1000) >= 1000 * 28
    assert lib.bsr_image_bytes(1920, 1080) >= 1920 * 1080 * 8 + 8160 * 4
    # include/bloomscene_anchors.h: one u32 per workgroup of 256 // K anchors, + the total
    assert lib.bsr_anchor_scratch_bytes(1000, 10) >= (1000 // 25 + 1) * 4
    assert lib.bsr_anchor_scratch_bytes(1000, 0) == 0 and lib.bsr_anchor_scratch_bytes(1000, 257) == 0


def test_every_entry_point_cites_the_reference_interface_it_replaces():
    hdr = open(os.path.join(ROOT, "include", "bloomscene_rast.h")).read()
    for fn, cite in [("bsr_forward", "rasterizer.h:31-54"), ("bsr_backward", "rasterizer.h:75-105"),
                     ("bsr_visible_filter", "rasterizer.h:57-73"), ("bsr_mark_visible", "rasterizer.h:24-29")]:
        assert cite in hdr, (fn, cite)


def test_settings_tuple_matches_reference_fields():
    # depth_diff_gaussian_rasterization/__init__.py:158-170
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")


def test_shim_import_path_resolves_to_this_package():
    # reference gaussian_renderer/__init__.py:16
    from depth_diff_gaussian_rasterization import GaussianRasterizationSettings as S, GaussianRasterizer as R
    assert S is GaussianRasterizationSettings and R is GaussianRasterizer


def _settings():
    cam = cameras.identity_camera(32, 32, math.radians(60))
    return views.make_settings(cam, torch.zeros(3), 1)


def test_argument_validation_matches_reference_messages():
    r = GaussianRasterizer(_settings())
    m = torch.zeros(4, 3)
    o = torch.ones(4, 1)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o, scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, o, shs=torch.zeros(4, 1, 3), colors_precomp=torch.zeros(4, 3), scales=torch.ones(4, 3),
          rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(m, m, o, colors_precomp=torch.zeros(4, 3), scales=torch.ones(4, 3))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(m, m, o, colors_precomp=torch.zeros(4, 3), scales=torch.ones(4, 3), rotations=torch.ones(4, 4),
          cov3D_precomp=torch.zeros(4, 6))


def test_no_cpu_fallback():
    """CPU tensors must fail loudly -- the product never routes through a CPU path or the oracle."""
    r = GaussianRasterizer(_settings())
    m = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(m, m, torch.ones(4, 1), colors_precomp=torch.zeros(4, 3), scales=torch.ones(4, 3),
          rotations=torch.ones(4, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        r.visible_filter(m, torch.ones(4, 3), torch.ones(4, 4))
    with pytest.raises(RuntimeError, match="no CPU path"):
        r.markVisible(m)
    with pytest.raises(RuntimeError, match="num_points, 3"):   # rasterize_points.cu:57-59
        r(torch.zeros(4, 2), m, torch.ones(4, 1), colors_precomp=torch.zeros(4, 3), scales=torch.ones(4, 3),
          rotations=torch.ones(4, 4))
    import bloomscene_amd
    src = "".join(open(os.path.join(os.path.dirname(bloomscene_amd.__file__), f)).read()
                  for f in os.listdir(os.path.dirname(bloomscene_amd.__file__)) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_camera_conventions():
    """A0: viewmatrix = W2C^T, projmatrix = viewmatrix @ P^T, campos = camera position, checked against matrices
    worked out by hand: 8-view rotate360 sweep at 64x48, camera_angle_x = 60 deg -> FoVx = 57 deg
    (dataset_readers.py:105), tan(28.5 deg) = 0.5429557, tan(FoVy / 2) = 0.5429557 * 48 / 64 = 0.4072168;
    view 1 is yawed 45 deg about +Y (trajectory.py:16-24): W2C rotation [[c, 0, s], [0, 1, 0], [-s, 0, c]], stored
    transposed; znear 0.01, zfar 100: P[2][2] = 100 / 99.99, P[2][3] = -1 / 99.99, P[3][2] = 1."""
    W, H = 64, 48
    cams = cameras.rotate360_cameras(8, W, H, math.radians(60))
    assert len(cams) == 8
    r = 0.70710677
    want_view = {1: [[r, 0, -r, 0], [0, 1, 0, 0], [r, 0, r, 0], [0, 0, 0, 1]],
                 3: [[-r, 0, -r, 0], [0, 1, 0, 0], [r, 0, -r, 0], [0, 0, 0, 1]],
                 0: [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
                 4: [[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]]}
    fx, fy, a, b = 1.8417708, 2.4556944, 1.00010001, -0.010001
    want_proj = {1: [[fx * r, 0, -r * a, -r], [0, fy, 0, 0], [fx * r, 0, r * a, r], [0, 0, b, 0]],
                 3: [[-fx * r, 0, -r * a, -r], [0, fy, 0, 0], [fx * r, 0, -r * a, -r], [0, 0, b, 0]],
                 0: [[fx, 0, 0, 0], [0, fy, 0, 0], [0, 0, a, 1], [0, 0, b, 0]]}
    for i, m in want_view.items():
        assert torch.allclose(cams[i].world_view_transform, torch.tensor(m, dtype=torch.float32), atol=1e-7), i
    for i, m in want_proj.items():
        assert torch.allclose(cams[i].full_proj_transform, torch.tensor(m, dtype=torch.float32), atol=2e-6), i
    for i, cam in enumerate(cams):
        V = cam.world_view_transform.double()
        assert torch.allclose(V[:3, :3] @ V[:3, :3].T, torch.eye(3, dtype=torch.float64), atol=1e-6)
        assert torch.equal(cam.camera_center, torch.zeros(3))                 # the sweep never translates
        assert abs(cam.FoVx - math.radians(57.0)) < 1e-12 and abs(math.tan(cam.FoVy / 2) - 0.4072168) < 1e-7
        # a point on the optical axis of view i projects to the image centre with w = depth
        axis = V[:3, 2]          # third column of the stored matrix = camera z axis in world coords
        p = torch.cat([axis * 3.0, torch.ones(1, dtype=torch.float64)])
        hom = p @ cam.full_proj_transform.double()
        assert abs(hom[0] / hom[3]) < 1e-5 and abs(hom[1] / hom[3]) < 1e-5 and abs(hom[3] - 3.0) < 1e-5
    cam0 = views.yawed_camera(W, H, math.radians(60), 0.0)
    ident = cameras.identity_camera(W, H, math.radians(60))
    assert torch.equal(cam0.full_proj_transform, ident.full_proj_transform)
    # a translated camera: p_view = R^T p_world + t, campos = -R t; the world point campos maps to view-space 0
    Rm = cameras.yaw_rotation(30.0)
    cam = cameras.make_minicam(Rm, [0.5, -1.0, 2.0], 1.0, 0.8, 40, 30)
    c = cam.camera_center.double()
    assert torch.allclose(torch.cat([c, torch.ones(1, dtype=torch.float64)]) @ cam.world_view_transform.double(),
                          torch.tensor([0, 0, 0, 1.0], dtype=torch.float64), atol=1e-6)
    assert torch.allclose(c, torch.tensor(-(Rm @ [0.5, -1.0, 2.0])), atol=1e-6)
    # yawed_camera(+y): the optical axis turns towards +x
    z_axis = views.yawed_camera(W, H, 1.0, 20.0).world_view_transform[:3, 2]
    assert abs(z_axis[0].item() - math.sin(math.radians(20))) < 1e-6 and abs(z_axis[2].item() - math.cos(math.radians(20))) < 1e-6


def test_shard_views_round_robin_partition():
    for n, world in [(64, 1), (64, 2), (64, 8), (7, 4), (3, 8)]:
        seen = sorted(i for r in range(world) for i in views.shard_views(n, r, world))
        assert seen == list(range(n))
        sizes = [len(views.shard_views(n, r, world)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def _bitonic_asc(keys):
    """Python model of bitonic_sort_asc in bloomscene_amd/csrc/binning.hip (same index maths):
    all-ascending network, compare-exchanges with an upper index >= n are skipped."""
    k = list(keys)
    n = len(k)
    n2 = 1
    while n2 < n:
        n2 <<= 1
    size = 2
    while size <= n2:
        half = size >> 1
        for i in range(n2 >> 1):
            blk, off = divmod(i, half)
            lo, hi = blk * size + off, blk * size + size - 1 - off
            if hi < n and k[lo] > k[hi]:
                k[lo], k[hi] = k[hi], k[lo]
        stride = half >> 1
        while stride > 0:
            for i in range(n2 >> 1):
                lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1))
                hi = lo | stride
                if hi < n and k[lo] > k[hi]:
                    k[lo], k[hi] = k[hi], k[lo]
            stride >>= 1
        size <<= 1
    return k


def _bitonic_hybrid(keys, ch):
    """Python model of k_sort_tiles_global: chunks of `ch` sorted on their own, then every merge
    runs its far steps (partner distance >= ch) over the whole array and the near ones per chunk."""
    k = list(keys)
    n = len(k)
    n2 = 1
    while n2 < n:
        n2 <<= 1
    for base in range(0, n, ch):
        k[base:base + ch] = _bitonic_asc(k[base:base + ch])

    def stride_step(arr, m, pairs, stride):
        for i in range(pairs):
            lo = ((i & ~(stride - 1)) << 1) | (i & (stride - 1))
            hi = lo | stride
            if hi < m and arr[lo] > arr[hi]:
                arr[lo], arr[hi] = arr[hi], arr[lo]

    size = 2 * ch
    while size <= n2:
        half = size >> 1
        for i in range(n2 >> 1):
            blk, off = divmod(i, half)
            lo, hi = blk * size + off, blk * size + size - 1 - off
            if hi < n and k[lo] > k[hi]:
                k[lo], k[hi] = k[hi], k[lo]
        stride = size >> 2
        while stride >= ch:
            stride_step(k, n, n2 >> 1, stride)
            stride >>= 1
        for base in range(0, n, ch):
            part = k[base:base + ch]
            stride = ch >> 1
            while stride > 0:
                stride_step(part, len(part), ch >> 1, stride)
                stride >>= 1
            k[base:base + ch] = part
        size <<= 1
    return k


@pytest.mark.parametrize("n", [0, 1, 2, 3, 5, 8, 17, 100, 255, 256, 257, 1000, 1025])
def test_sort_network_sorts_any_length(n):
    rng = np.random.default_rng(n)
    keys = [int(x) for x in rng.integers(0, 50, size=n)]   # many duplicates
    keys = [(d << 32) | i for i, d in enumerate(keys)]       # (depth, id) keys are unique
    rng.shuffle(keys)
    assert _bitonic_asc(keys) == sorted(keys)
    for ch in (8, 64):
        if n > ch:
            assert _bitonic_hybrid(keys, ch) == sorted(keys)


def test_algorithmic_byte_model_is_consistent():
    """bench.py's per-stage bytes must add up to the contract figure of SURVEY.md §8d."""
    import bench
    P, M, R, N = 1_000_000, 16, 4_380_000, 1920 * 1080
    a = bench.algorithmic_bytes(P, M, R, N)
    fwd = a["preprocess"] + 8 * P + a["binning"] + a["sort_tiles"] + 8 * R + a["render_fwd"]
    assert abs(fwd - bench.step_bytes(P, M, R, N, False)) / fwd < 0.01
    total = fwd + a["render_bwd"] + a["preprocess_bwd"]
    assert abs(total - bench.step_bytes(P, M, R, N, True)) / total < 0.01
    assert abs(bench.step_bytes(P, M, R, N, True) - 1.40e9) / 1.40e9 < 0.01


def test_scratch_buffers_die_without_the_cycle_collector():
    """The resize callback must not form a reference cycle with its owner: a cycle kept three scratch tensors
    (hundreds of MB at C3) per call alive until the collector ran, i.e. a hipMalloc per step in steady state."""
    import gc
    import weakref
    from bloomscene_amd import rasterizer as RZ
    was = gc.isenabled()
    gc.disable()
    try:
        s = RZ._Scratch(torch.device("cpu"))
        assert s.callback(None, 4096) == s.tensor.data_ptr() and s.tensor.numel() == 4096
        w = weakref.ref(s.tensor)
        del s
        assert w() is None
    finally:
        if was:
            gc.enable()


def test_render_views_batched_argument_checks_need_no_gpu():
    """The batched sweep validates its cameras and colour inputs before it touches the device."""
    from bloomscene_amd.views import render_views_batched, yawed_camera
    g = dict(means3D=torch.zeros(4, 3), opacities=torch.ones(4, 1), colors_precomp=torch.zeros(4, 3),
             scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    bg = torch.zeros(3)
    with pytest.raises(ValueError, match="at least one camera"):
        render_views_batched([], g, bg, 0)
    a, b = yawed_camera(64, 48, 1.0, 0.0), yawed_camera(80, 48, 1.0, 5.0)
    with pytest.raises(ValueError, match="one image size and field of view"):
        render_views_batched([a, b], g, bg, 0)
    both = dict(g, shs=torch.zeros(4, 1, 3))
    with pytest.raises(Exception, match="exactly one|excatly one"):
        render_views_batched([a], both, bg, 0)
    with pytest.raises(RuntimeError, match="GPU tensor"):   # CPU tensors: there is no CPU path
        render_views_batched([a], g, bg, 0)


def test_camera_pack_and_multi_view_helpers_on_cpu():
    """views.CameraPack stacks a path's matrices once; select() slices consecutive views and gathers others; the
    fused-front-end and option entry points fail loudly without a GPU (no CPU path)."""
    from bloomscene_amd import _capi, views
    from bloomscene_amd.cameras import rotate360_cameras
    cams = rotate360_cameras(12, 64, 48, 1.0)
    pack = views.CameraPack(cams, "cpu")
    assert len(pack) == 12 and pack.world_view.shape == (12, 4, 4) and pack.centers.shape == (12, 3)
    vms, pms, cps = pack.select([3, 4, 5])
    assert torch.equal(vms, torch.stack([cams[i].world_view_transform for i in (3, 4, 5)]))
    vms, pms, cps = pack.select([7, 2])
    assert torch.equal(pms, torch.stack([cams[i].full_proj_transform for i in (7, 2)]))
    with pytest.raises(ValueError, match="one image size"):
        views.CameraPack([cams[0], views.yawed_camera(80, 48, 1.0, 0.0)], "cpu")
    with pytest.raises(RuntimeError, match="GPU"):
        views.group_visibility(pack, torch.zeros(4, 3), torch.ones(4, 3), torch.ones(4, 4), [[0, 1], [2]])
    # the row gathers of the sweep paths: argument checks before anything native, and no CPU path either
    from bloomscene_amd import rasterizer as RZ
    with pytest.raises(RuntimeError, match="GPU"):
        RZ._gather_rows_native([torch.zeros(4, 3)], torch.zeros(2, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="1..8"):
        RZ._gather_rows_native([], torch.zeros(2, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="GPU"):
        views.compact_for_view_groups(pack, dict(means3D=torch.zeros(4, 3), scales=torch.ones(4, 3),
                                                 rotations=torch.ones(4, 4)), [[0], [1]])
    from bloomscene_amd.neural_gaussians import render_anchors
    with pytest.raises(RuntimeError, match="GPU"):
        render_anchors(torch.zeros(2, 3), torch.ones(2, 6), torch.zeros(2, 5, 3), torch.ones(10, 1), torch.zeros(10, 3),
                       torch.zeros(10, 7), _settings())
    # the library keeps no switch between calls: the former process-wide test hooks are per-call flags now (round 6),
    # and the python context that carries them is thread-local and nestable
    assert "bsr_set_option" not in _capi.SIGNATURES and not hasattr(_capi, "set_option")
    from bloomscene_amd import numerics
    from bloomscene_amd.numerics import FLAG_TEST_NO_HALF_MASKS, FLAG_TEST_SORT_INT, resolve_flags
    assert resolve_flags() == 0
    with numerics(exact_exp=True, test_flags=FLAG_TEST_SORT_INT):
        assert resolve_flags() == 1 | FLAG_TEST_SORT_INT
        with numerics(test_flags=FLAG_TEST_NO_HALF_MASKS):        # replaces the enclosing block's test bits, keeps its numerics
            assert resolve_flags() == 1 | FLAG_TEST_NO_HALF_MASKS
        with numerics(strict_gradients=True):                      # None keeps the test bits
            assert resolve_flags(exact_exp=False) == 2 | FLAG_TEST_SORT_INT
        assert resolve_flags() == 1 | FLAG_TEST_SORT_INT
    assert resolve_flags() == 0


def test_bench_refuses_more_ranks_than_gpus_with_a_message():
    """`python bench.py --gpus N` starts its own ranks; where the box shows fewer GPUs than ranks it says so (rc != 0)
    instead of failing inside a child."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "BSR_BENCH_SINGLE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "GPU(s)" in (r.stderr + r.stdout)


def test_numerics_context_is_thread_local_and_nests():
    """The python host's default mode of a call (bloomscene_amd.numerics) is per THREAD and nests; explicit arguments
    win over it.  (The C library itself holds no numerics state: include/bloomscene_rast.h BSR_FLAG_*.)"""
    import threading
    from bloomscene_amd import numerics
    from bloomscene_amd.numerics import FLAG_EXACT_EXP, FLAG_EXACT_GRAD, resolve_flags
    assert resolve_flags() == 0
    seen = {}
    with numerics(exact_exp=True):
        assert resolve_flags() == FLAG_EXACT_EXP
        with numerics(strict_gradients=True):
            assert resolve_flags() == FLAG_EXACT_EXP | FLAG_EXACT_GRAD
            assert resolve_flags(exact_exp=False) == FLAG_EXACT_GRAD
            with numerics(exact_exp=False, strict_gradients=False):
                assert resolve_flags() == 0
            t = threading.Thread(target=lambda: seen.setdefault("other", resolve_flags()))
            t.start()
            t.join()
        assert resolve_flags() == FLAG_EXACT_EXP
        assert resolve_flags(strict_gradients=True) == FLAG_EXACT_EXP | FLAG_EXACT_GRAD
    assert resolve_flags() == 0 and seen["other"] == 0
    from bloomscene_amd import GaussianRasterizer
    r = GaussianRasterizer(_settings(), exact_exp=True, strict_gradients=True)
    assert r._flags() == 3 and GaussianRasterizer(_settings())._flags() == 0
