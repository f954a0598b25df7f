"""GPU parity tests of the fused anchor expansion (bloomscene_amd.neural_gaussians, C ABI of
include/bloomscene_anchors.h) against the oracle of reference gaussian_renderer/__init__.py:165-203
(oracle/anchors.py on CPU, fp32 forward / float64 autograd for gradients) and against the same torch
ops run eagerly on the GPU, which is what the reference executes on this hardware.

Tolerances: selection mask, order, opacity, colour and xyz (one multiply + one add, same roundings
as torch) are bit-exact; scaling (sigmoid: exp differs by <= 1 ulp between libraries) and rot
(norm summation order) within 4 ulp-ish 1e-6 relative; gradients <= 1e-5 of the tensor's scale vs the
float64 oracle (fp32 arithmetic, sums of <= K terms)."""
import numpy as np
import pytest
import torch

from oracle import anchors as OA

pytestmark = pytest.mark.gpu

NAMES = ("xyz", "color", "opacity", "scaling", "rot")
IN_NAMES = ("anchor", "grid_scaling", "grid_offsets", "neural_opacity", "color", "scale_rot")


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda")


def _hip(inp, need_grad=False):
    from bloomscene_amd.neural_gaussians import expand_anchors
    dev = _dev()
    leaves = [t.to(dev).requires_grad_(need_grad) for t in inp]
    return leaves, expand_anchors(*leaves)


def _check_forward(inp):
    ref = OA.expand_anchors_reference(*inp)
    _, out = _hip(inp)
    torch.cuda.synchronize()
    assert out[5].dtype == torch.bool
    assert torch.equal(out[5].cpu(), ref[5])
    for name, a, b in zip(NAMES, out[:5], ref[:5]):
        a = a.cpu()
        assert a.shape == b.shape and a.dtype == torch.float32, name
        if name in ("xyz", "color", "opacity"):
            assert torch.equal(a.view(torch.int32), b.view(torch.int32)), name
        else:
            err = (a.double() - b.double()).abs()
            assert (err <= 1e-6 * b.double().abs() + 1e-30).all(), (name, float((err / (b.abs() + 1e-30)).max()))
    return ref, out


@pytest.mark.parametrize("N,K,seed,keep", [(1000, 10, 0, 0.5), (777, 5, 1, 0.3), (4099, 1, 2, 0.7), (5, 256, 3, 0.5),
                                           (333, 7, 4, 0.5), (2, 10, 5, 0.5), (20000, 10, 6, 0.45)])
def test_forward_vs_oracle(N, K, seed, keep):
    _check_forward(OA.synthetic_anchor_inputs(N, K, seed=seed, keep_fraction=keep, zero_quat_rows=min(3, N)))


def test_forward_vs_torch_eager_on_gpu():
    inp = OA.synthetic_anchor_inputs(5000, 10, seed=7)
    dev = _dev()
    ref = OA.expand_anchors_reference(*[t.to(dev) for t in inp])
    _, out = _hip(inp)
    assert torch.equal(out[5], ref[5])
    for name, a, b in zip(NAMES, out[:5], ref[:5]):
        if name in ("xyz", "color", "opacity"):
            assert torch.equal(a, b), name
        else:
            assert ((a - b).abs() <= 1e-6 * b.abs()).all(), name


def test_edge_selections():
    dev = _dev()
    for N, K in [(0, 10), (64, 10)]:
        for mode in ("none", "all"):
            inp = list(OA.synthetic_anchor_inputs(max(N, 1), K, seed=8))
            if N == 0:
                inp = [t[:0] for t in inp]
            inp[3] = torch.full_like(inp[3], -1.0 if mode == "none" else 0.25)
            ref, out = _check_forward(inp)
            assert out[0].shape[0] == (0 if mode == "none" else N * K)
    # zeros are NOT selected (strict > 0, GR:169); negative zero neither; NaN neither
    inp = list(OA.synthetic_anchor_inputs(10, 10, seed=9))
    inp[3][:50] = 0.0
    inp[3][50:60] = -0.0
    inp[3][60:65] = float("nan")
    ref, out = _check_forward(inp)
    assert not out[5][:65].any()


def _grads(inp, upstream, fn, dtype, dev):
    leaves = [t.to(dev, dtype).requires_grad_(True) for t in inp]
    out = fn(*leaves)
    torch.autograd.backward(list(out[:5]), [u.to(dev, dtype) for u in upstream])
    return [l.grad.detach().cpu().double() for l in leaves]


@pytest.mark.parametrize("N,K,seed", [(1000, 10, 0), (777, 5, 1), (4099, 1, 2), (5, 256, 3), (333, 7, 4)])
def test_backward_vs_float64_autograd(N, K, seed):
    from bloomscene_amd.neural_gaussians import expand_anchors
    inp = OA.synthetic_anchor_inputs(N, K, seed=seed)
    S = int((inp[3] > 0).sum())
    g = torch.Generator().manual_seed(seed + 100)
    upstream = [torch.randn(S, w, generator=g) for w in (3, 3, 1, 3, 4)]
    ref = _grads(inp, upstream, OA.expand_anchors_reference, torch.float64, "cpu")
    got = _grads(inp, upstream, expand_anchors, torch.float32, _dev())
    for name, a, b in zip(IN_NAMES, got, ref):
        assert a.shape == b.shape, name
        assert torch.isfinite(a).all(), name
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 1e-5 * scale, (name, float((a - b).abs().max()) / scale)
    sel = (inp[3] > 0).view(-1)
    for idx in (2, 4, 5):   # per-candidate gradients of unselected rows are exactly zero
        assert not got[idx].reshape(N * K, -1)[~sel].any()
    assert not got[3].reshape(-1)[~sel].any()


def test_backward_zero_quaternion_and_missing_upstreams():
    from bloomscene_amd.neural_gaussians import expand_anchors
    dev = _dev()
    inp = list(OA.synthetic_anchor_inputs(50, 10, seed=11))
    inp[3] = inp[3].abs() + 0.1
    inp[5][:7, 3:7] = 0.0
    leaves = [t.to(dev).requires_grad_(True) for t in inp]
    out = expand_anchors(*leaves)
    assert not out[4][:7].any()          # 0 / max(0, 1e-12)
    out[3].sum().backward()              # only `scaling` carries a gradient: the other upstreams are absent
    ref_leaves = [t.double().requires_grad_(True) for t in inp]
    ref = OA.expand_anchors_reference(*ref_leaves)
    ref[3].sum().backward()
    for name, a, b in zip(IN_NAMES, leaves, ref_leaves):
        got = a.grad.cpu().double()
        want = b.grad if b.grad is not None else torch.zeros_like(b)   # torch: no path -> no gradient
        assert float((got - want).abs().max()) <= 1e-5 * (float(want.abs().max()) + 1e-30), name
    assert not leaves[0].grad.any() and not leaves[2].grad.any() and not leaves[4].grad.any()


def test_backward_is_bit_reproducible_and_feeds_the_rasterizer():
    """expand_anchors -> GaussianRasterizer with colors_precomp, the reference's render() call shape
    (GR:254-262); gradients reach the six MLP-side tensors and are identical run to run."""
    import math
    from bloomscene_amd import GaussianRasterizer, cameras, views
    from bloomscene_amd.neural_gaussians import expand_anchors
    dev = _dev()
    W, H = 160, 96
    cam = cameras.identity_camera(W, H, math.radians(60)).to(dev)
    inp = list(OA.synthetic_anchor_inputs(3000, 10, seed=12))
    inp[0] = inp[0] * torch.tensor([0.6, 0.35, 0.0]) + torch.tensor([0.0, 0.0, 6.0])   # in front of the camera
    gC = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)

    def run(fn):
        leaves = [t.to(dev).requires_grad_(True) for t in inp]
        xyz, color, opacity, scaling, rot, mask = fn(*leaves)
        means2D = torch.zeros_like(xyz, requires_grad=True)
        rast = GaussianRasterizer(views.make_settings(cam, torch.zeros(3, device=dev), 1))
        img, radii, depth = rast(means3D=xyz, means2D=means2D, opacities=opacity, colors_precomp=color,
                                 scales=scaling, rotations=rot)
        (img * gC).sum().backward()
        return img.detach(), [l.grad.clone() for l in leaves], int((radii > 0).sum())

    img1, g1, vis = run(expand_anchors)
    img2, g2, _ = run(expand_anchors)
    assert vis > 1000
    assert torch.equal(img1, img2)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    # same pipeline with the torch-eager expansion in front of the same rasterizer
    img3, g3, _ = run(OA.expand_anchors_reference)
    assert float((img1 - img3).abs().max()) <= 1e-5
    for name, a, b in zip(IN_NAMES, g1, g3):
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 1e-4 * scale, (name, float((a - b).abs().max()) / scale)


def test_render_neural_result_dict():
    """views.render_neural: the training-mode result dict of the reference's render() (GR:266-279) for the
    anchor representation; viewspace_points receives the screen-space gradient after retain_grad()."""
    import math
    from bloomscene_amd import cameras, views
    dev = _dev()
    W, H = 128, 80
    cam = cameras.identity_camera(W, H, math.radians(60)).to(dev)
    inp = list(OA.synthetic_anchor_inputs(2000, 10, seed=21))
    inp[0] = inp[0] * torch.tensor([0.6, 0.35, 0.0]) + torch.tensor([0.0, 0.0, 6.0])
    leaves = [t.to(dev).requires_grad_(True) for t in inp]
    res = views.render_neural(cam, *leaves, torch.zeros(3, device=dev), retain_grad=True)
    assert set(res) == {"render", "viewspace_points", "visibility_filter", "radii", "depth", "selection_mask",
                        "neural_opacity", "scaling"}
    S = int((inp[3] > 0).sum())
    assert res["render"].shape == (3, H, W) and res["depth"].shape == (1, H, W)
    assert res["radii"].shape == (S,) and res["visibility_filter"].dtype == torch.bool
    assert res["selection_mask"].shape == (20000,) and int(res["selection_mask"].sum()) == S
    assert res["scaling"].shape == (S, 3) and res["neural_opacity"] is leaves[3]
    (res["render"].sum() + res["scaling"].prod(dim=1).mean()).backward()      # image loss + scaling regulariser
    vp = res["viewspace_points"].grad
    assert vp is not None and vp.shape == (S, 3) and bool(vp[res["visibility_filter"]].abs().sum() > 0)
    assert not vp[:, 2].any()
    for leaf in leaves:
        assert leaf.grad is not None and torch.isfinite(leaf.grad).all()


@pytest.mark.parametrize("depth_gradient", [False, True])
def test_fused_native_render_equals_the_two_autograd_nodes(depth_gradient):
    """neural_gaussians.render_anchors = bsr_anchor_render_forward / _backward: selection + expansion + rasterizer in
    one native call each way (no interpreter time between the blocking read of the selection count and the
    rasterizer's first launch).  Image, depth, radii, mask, the expanded tensors and every gradient -- with a loss that
    also reads `scaling` and `opacity` directly, as BloomScene's regularisers do -- must be bit-identical to
    expand_anchors followed by GaussianRasterizer, and viewspace_points.grad must be the rasterizer's dL_dmean2D.
    Also: nothing selected, and no anchors at all."""
    import math
    from bloomscene_amd import cameras, views
    from bloomscene_amd.synthetic import upstream_grads
    dev = _dev()
    W, H = 160, 96
    cam = cameras.identity_camera(W, H, math.radians(60)).to(dev)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    gC, gD = upstream_grads(W, H, seed=3)
    gC, gD = gC.to(dev), gD.to(dev)
    for N, K, seed in ((3001, 10, 31), (500, 7, 32)):    # (odd selection counts: the packed sections stay aligned)
        inp = list(OA.synthetic_anchor_inputs(N, K, seed=seed))
        inp[0] = inp[0] * torch.tensor([0.6, 0.35, 0.0]) + torch.tensor([0.0, 0.0, 6.0])
        results = []
        for fused in (True, False):
            leaves = [t.to(dev).clone().requires_grad_(True) for t in inp]
            res = views.render_neural(cam, *leaves, bg, depth_gradient=depth_gradient, fused=fused)
            loss_extra = res["scaling"].prod(dim=1).mean() * 3.0
            torch.autograd.backward((res["render"], res["depth"], loss_extra), (gC, gD, torch.ones((), device=dev)))
            results.append((res, [t.grad.clone() for t in leaves], res["viewspace_points"].grad.clone()))
        (ra, ga, va), (rb, gb, vb) = results
        for k in ("render", "depth", "radii", "selection_mask", "scaling", "visibility_filter"):
            assert torch.equal(ra[k], rb[k]), k
        assert int(ra["radii"].numel()) % 2 == 1 or N == 500
        for name, x, y in zip(IN_NAMES, ga, gb):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32)), name
        assert torch.equal(va.view(torch.int32), vb.view(torch.int32)) and bool(va.abs().sum() > 0)
    # nothing selected / no anchors: zero images, empty per-Gaussian outputs, zero gradients
    for N, K in ((40, 10), (0, 10)):
        inp = list(OA.synthetic_anchor_inputs(max(N, 1), K, seed=1))
        inp = [t[:N] if i < 3 else t[:N * K] for i, t in enumerate(inp)]
        inp[3] = -inp[3].abs() - 1.0
        leaves = [t.to(dev).clone().requires_grad_(True) for t in inp]
        res = views.render_neural(cam, *leaves, bg, fused=True)
        assert res["radii"].numel() == 0 and res["scaling"].shape == (0, 3) and not res["selection_mask"].any()
        assert not res["render"].any() and not res["depth"].any()   # P == 0: zero images (rasterize_points.cu:68-82)
        res["render"].sum().backward()
        for leaf in leaves:
            assert leaf.grad is not None and not leaf.grad.any()


def test_training_view_equals_the_separate_calls():
    """SURVEY.md §8f rank 2, per-iteration half (bloomscene.py:240-243): views.training_view -- one rasterizer object,
    bsr_visible_filter_indices, index gathers, fused expansion, render -- against the reference's shape of the same
    iteration: prefilter_voxel (visible_filter > 0), six boolean indexes (GR:33-43), expansion, render.  The filter
    radii, the index list, the image, depth, radii and every gradient w.r.t. the FULL per-anchor parameters must be
    bit-identical; the index list must be what nonzero() gives."""
    from bloomscene_amd import GaussianRasterizer, views
    from bloomscene_amd.synthetic import anchor_scene, upstream_grads
    dev = torch.device("cuda")
    N, K, W, H = 30000, 10, 320, 200
    sc = anchor_scene(N, K, W, H, seed=5)
    g = torch.Generator().manual_seed(9)
    # half the anchors behind / beside the camera, so the filter really filters
    sc.anchor[torch.randperm(N, generator=g)[: N // 2], 2] *= -1.0
    rot = torch.nn.functional.normalize(torch.randn(N, 4, generator=g))
    cam = sc.camera.to(dev)
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    gC, gD = upstream_grads(W, H, seed=2)
    gC, gD = gC.to(dev), gD.to(dev)
    heads_full = tuple(t.to(dev) for t in (sc.neural_opacity.view(N, K, 1), sc.color.view(N, K, 3), sc.scale_rot.view(N, K, 7)))

    def leaves():
        return [t.to(dev).clone().requires_grad_(True) for t in (sc.anchor, sc.grid_scaling, sc.grid_offsets)] + \
               [t.clone().requires_grad_(True) for t in heads_full]

    # ---- the fused per-iteration path
    a, s6, off, h_op, h_col, h_sr = leaves()

    def heads(idx):
        return (h_op.index_select(0, idx).reshape(-1, 1), h_col.index_select(0, idx).reshape(-1, 3),
                h_sr.index_select(0, idx).reshape(-1, 7))
    res = views.training_view(cam, a, s6, rot.to(dev), off, heads, bg)
    torch.autograd.backward((res["render"], res["depth"]), (gC, gD))
    got = [t.grad.clone() for t in (a, s6, off, h_op, h_col, h_sr)]

    # ---- the reference's shape of the same iteration
    a2, s62, off2, h_op2, h_col2, h_sr2 = leaves()
    mask = views.prefilter(cam, a2, s62, rot.to(dev), bg)
    assert 0.1 * N < int(mask.sum()) < 0.9 * N
    va, vs, vo = a2[mask], s62[mask], off2[mask]                       # GR:33-38
    ref = views.render_neural(cam, va, vs, vo, h_op2[mask].reshape(-1, 1), h_col2[mask].reshape(-1, 3),
                              h_sr2[mask].reshape(-1, 7), bg)
    torch.autograd.backward((ref["render"], ref["depth"]), (gC, gD))
    want = [t.grad for t in (a2, s62, off2, h_op2, h_col2, h_sr2)]

    assert torch.equal(res["visible_mask"], mask)
    assert torch.equal(res["visible_idx"], mask.nonzero().squeeze(1))
    rast = GaussianRasterizer(views.make_settings(cam, bg, 1))
    assert torch.equal(res["anchor_radii"], rast.visible_filter(a2.detach(), s62.detach()[:, :3], rot.to(dev)))
    for k in ("render", "depth", "radii", "selection_mask"):
        assert torch.equal(res[k], ref[k]), k
    for name, x, y in zip(("anchor", "scaling", "offsets", "opacity head", "colour head", "scale_rot head"), got, want):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32)), name
    assert float(got[0].abs().sum()) > 0 and not got[0][~mask].any()      # invisible anchors: zero gradient rows
    # edge cases of the index entry point: nothing visible, everything visible, no points
    behind = sc.anchor.to(dev).clone()
    behind[:, 2] = -1.0
    r0, i0 = rast.visible_filter_indices(behind, sc.grid_scaling.to(dev)[:, :3], rot.to(dev))
    assert i0.numel() == 0 and not r0.any()
    front = sc.anchor.to(dev).clone()
    front[:, 2] = front[:, 2].abs()
    r1, i1 = rast.visible_filter_indices(front, sc.grid_scaling.to(dev)[:, :3], rot.to(dev))
    assert torch.equal(i1, (r1 > 0).nonzero().squeeze(1)) and i1.numel() > 0.9 * N
    r2, i2 = rast.visible_filter_indices(front[:0], sc.grid_scaling.to(dev)[:0, :3], rot.to(dev)[:0])
    assert r2.numel() == 0 and i2.numel() == 0
