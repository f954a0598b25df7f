import sys, torch, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from bloomscene_amd import views, GaussianRasterizer
from bloomscene_amd.rasterizer import check_deferred
from bloomscene_amd.synthetic import anchor_scene, upstream_grads
which = sys.argv[1]
dev = torch.device('cuda')
N, K, W, H = 20000, 10, 256, 192
sc = anchor_scene(N, K, W, H, seed=3)
cam = sc.camera.to(dev); bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
names = ("anchor", "grid_scaling", "grid_offsets", "neural_opacity", "color", "scale_rot")
leaves = {k: getattr(sc, k).to(dev).requires_grad_(True) for k in names}
gC, gD = [t.to(dev) for t in upstream_grads(W, H, seed=1)]
settings = views.make_settings(cam, bg, 1)
cap = 4_000_000
def step():
    for v in leaves.values(): v.grad = None
    if which in ('A', 'B'):
        res = views.render_neural(cam, *[leaves[k] for k in names], bg, capacity=cap, settings=settings)
        if which == 'B':
            torch.autograd.backward((res["render"], res["depth"]), (gC, gD))
        return res
    if which in ('C', 'D'):
        P = N * K
        g = torch.Generator().manual_seed(0)
        if not hasattr(step, 't'):
            from bloomscene_amd.synthetic import scene_a
            s = scene_a(P, W, H, 0, seed=0)
            step.t = {k: getattr(s, k).to(dev).requires_grad_(True) for k in ("means3D", "scales", "rotations", "opacities")}
            step.col = torch.rand(P, 3).to(dev).requires_grad_(True)
            step.r = GaussianRasterizer(settings, capacity=cap)
        t = step.t
        m2d = torch.zeros_like(t["means3D"], requires_grad=True)
        color, radii, depth = step.r(means3D=t["means3D"], means2D=m2d, opacities=t["opacities"], colors_precomp=step.col, scales=t["scales"], rotations=t["rotations"])
        if which == 'D':
            torch.autograd.backward((color, depth), (gC, gD))
        return color
if which in ('E', 'F'):
    pre = views.render_neural(cam, *[leaves[k] for k in names], bg)
    torch.autograd.backward((pre["render"], pre["depth"]), (gC, gD))
    pre_grad = pre["viewspace_points"].grad.clone()
    if which == 'F':
        pre2 = views.render_neural(cam, *[leaves[k] for k in names], bg, capacity=cap)
        torch.autograd.backward((pre2["render"], pre2["depth"]), (gC, gD))
        check_deferred()
    which = 'B'
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize(); check_deferred()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
g.replay(); torch.cuda.synchronize()
print(which, "ok", flush=True)
