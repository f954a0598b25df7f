import sys, os, time, math
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as Hh
from bloomscene_amd import _capi
# clustered: all Gaussians within a small screen patch, mixed sizes incl. very large ones
c = Hh.make_case(P=150000, W=640, H=360, deg=1, seed=5, scale_mul=3.0)
g = torch.Generator().manual_seed(9)
c.means3D[:, 0] *= 0.08; c.means3D[:, 1] *= 0.08           # squeeze into the image centre
big = torch.randperm(c.P, generator=g)[:300]
c.scales[big] *= 60.0                                       # some splats covering hundreds of tiles
t=time.time(); st, gr = Hh.run_oracle(c); print("oracle %.1fs R=%d max list=%d" % (time.time()-t, st.num_rendered, (st.ranges[:,1]-st.ranges[:,0]).max()))
_capi.profile_enable(True)
t=time.time(); out = Hh.run_hip(c); torch.cuda.synchronize(); print("hip first call %.3fs" % (time.time()-t))
_capi.profile_reset()
t=time.time(); out = Hh.run_hip(c); torch.cuda.synchronize(); print("hip second call %.3fs" % (time.time()-t))
print({k: round(v[0]/max(v[1],1),3) for k,v in _capi.profile_read().items()})
print("radii eq", (out.radii==st.radii).all(), "color bit-eq", (out.color.view(np.uint32)==st.color.view(np.uint32)).all(), "depth bit-eq", (out.depth.view(np.uint32)==st.depth.view(np.uint32)).all())
og = Hh.oracle_grads(c, gr)
for k in ("means3D","means2D","opacities","shs","scales","rotations"):
    print(k, "err/scale %.2e" % Hh.max_err_over_scale(getattr(out.grads,k), getattr(og,k)))
