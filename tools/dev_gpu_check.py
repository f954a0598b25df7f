"""Developer script: detailed HIP-vs-oracle comparison on the GPU box (prints, asserts nothing)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import helpers as Hh
from bloomscene_amd import rasterizer as RZ

def stats(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    if a.dtype.kind == 'f':
        neq = int((a.view(np.uint32) != b.view(np.uint32)).sum()) if a.dtype == np.float32 and b.dtype == np.float32 else -1
        m, frac = Hh.rel_err(a, b)
        print(f"   {name:14s} bit-neq {neq:8d}/{a.size:9d}  max_rel {m:.3e}  frac>1e-4 {frac:.2e}  err/scale {Hh.max_err_over_scale(a,b):.3e}")
    else:
        print(f"   {name:14s} neq {int((a != b).sum())}/{a.size}")

cases = [
    dict(P=2000, W=160, H=96, deg=3),
    dict(P=3000, W=133, H=75, deg=1, scale_mul=4.0, near_fraction=0.1),
    dict(P=1500, W=64, H=64, deg=0, color_mode="precomp"),
    dict(P=1500, W=100, H=50, deg=2, cov_mode="precomp", scale_mul=3.0),
    dict(P=4000, W=48, H=48, deg=3, scale_mul=10.0),   # long per-tile lists (> 1024)
    dict(P=100000, W=800, H=800, deg=1),
]
for kw in cases:
    c = Hh.make_case(**kw)
    t = time.time(); st, g = Hh.run_oracle(c); to = time.time() - t
    # raw native call to also fetch buffers
    out = Hh.run_hip(c)
    print(kw, "R", st.num_rendered, f"oracle {to:.2f}s")
    stats("radii", out.radii, st.radii)
    stats("color", out.color, st.color)
    stats("depth", out.depth, st.depth)
    og = Hh.oracle_grads(c, g)
    for k in ("means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"):
        a, b = getattr(out.grads, k), getattr(og, k)
        if b is None:
            continue
        if a is None:
            print("   MISSING grad", k); continue
        stats("d_" + k, a.reshape(-1), b.reshape(-1))
    # internal state
    dev = torch.device("cuda")
    rs = Hh.hip_settings(c, dev)
    e = torch.Tensor([])
    def d(t): return e if t is None else t.to(dev)
    nr, col, dep, rad, gb, bb, ib = RZ._rasterize_gaussians_native(rs.bg, c.means3D.to(dev), d(c.colors_precomp), c.opacities.to(dev), d(c.scales), d(c.rotations), rs.scale_modifier, d(c.cov3D_precomp), rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, c.H, c.W, d(c.shs), c.deg, rs.campos, False, True)
    torch.cuda.synchronize()
    b = Hh.decode_buffers(c.P, c.W, c.H, nr, gb, bb, ib)
    print("   num_rendered", nr, st.num_rendered)
    vis = st.radii > 0
    stats("xy", b.rec[vis][:, 0:2], st.means2D[vis])
    stats("conic", np.concatenate([b.rec[vis][:, 2:4], b.rec[vis][:, 4:5]], 1), st.conic_opacity[vis][:, :3])
    stats("depths", b.rec[vis][:, 7], st.depths[vis])
    feat = st.features if st.features is not None else st.rgb
    stats("rgb", b.rec[vis][:, 8:11], np.asarray(feat)[vis])
    stats("cov3D", b.cov3D[vis], (st.cov3D if c.cov3D_precomp is None else c.cov3D_precomp.numpy())[vis])
    stats("point_list", b.point_list, st.point_list)
    stats("ranges", b.tile_start[:-1], st.ranges[:, 0] if nr else b.tile_start[:-1])
    stats("final_T", b.final_T, st.final_T)
    stats("n_contrib", b.n_contrib, st.n_contrib)
