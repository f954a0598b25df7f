#!/usr/bin/env python3
"""SURVEY.md §8(d) parity metric of the HIP path against the CPU oracle, per tensor:
    max over elements of |a - b| / max(|b|, 1e-6 * max|b|)   and the fraction of elements above 1e-4
for colour, depth and every gradient, plus the norm-wise max|a - b| / max|b|.  Runs on the GPU box:

    python tools/parity_report.py [--lib build.so] [--exact-exp] [c1 c2 c3 c5 dense ...]  > profiles/<round>/parity_report.jsonl

--lib: another build of the same C ABI (the attribution builds of csrc/Makefile); --exact-exp: the forward with the
pinned exp everywhere (bit-equal to the oracle) instead of the library's default.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as Hh  # noqa: E402

CASES = {
    "c1": dict(P=10_000, W=256, H=256, deg=0, seed=0),
    "c2": dict(P=100_000, W=800, H=800, deg=1, seed=0),
    "c3": dict(P=1_000_000, W=1920, H=1080, deg=3, seed=0),
    "c5": dict(P=5_000_000, W=1920, H=1080, deg=3, seed=0),
    "dense": dict(P=200_000, W=960, H=540, deg=3, seed=4, scale_mul=4.0),
    "free_camera": dict(P=100_000, W=640, H=360, deg=2, seed=13, scale_mul=3.0, free_camera=True),
    "precomp": dict(P=200_000, W=512, H=512, deg=1, seed=2, color_mode="precomp", scale_mul=2.0),
    "lists": dict(P=20_000, W=48, H=48, deg=1, seed=7, scale_mul=12.0),   # ~2000 entries per tile: long T chains
}


GRADS = ("means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp")


def main():
    argv = sys.argv[1:]
    lib = None
    exact = False
    while argv and argv[0].startswith("--"):
        if argv[0] == "--lib":
            lib, argv = argv[1], argv[2:]
        elif argv[0] == "--exact-exp":
            exact, argv = True, argv[1:]
        else:
            raise SystemExit(f"unknown option {argv[0]}")
    from bloomscene_amd import _capi
    if lib:
        _capi.use_library(lib)
    from bloomscene_amd import numerics
    numerics(exact_exp=exact).__enter__()   # per-call flags of every call this process makes
    names = argv or ["c1", "c2", "c3"]
    for name in names:
        c = Hh.make_case(**CASES[name])
        t0 = time.time()
        st, g = Hh.run_oracle(c)
        t_or = time.time() - t0
        out = Hh.run_hip(c)
        # the strict mode beside the default (BSR_FLAG_EXACT_GRAD: the reference's per-pair operations in the backward walk)
        out_strict = Hh.run_hip(c, strict_gradients=True)
        rec = {"case": name, **{k: v for k, v in CASES[name].items()}, "num_rendered": int(st.num_rendered),
               "oracle_s": round(t_or, 1), "lib": os.path.basename(lib) if lib else "product",
               "exact_exp": bool(exact),
               "radii_equal": bool((out.radii == st.radii).all()),
               "color_bit_exact": bool((out.color.view(np.uint32) == st.color.view(np.uint32)).all()),
               "depth_bit_exact": bool((out.depth.view(np.uint32) == st.depth.view(np.uint32)).all()),
               "tensors": {}}
        og = Hh.oracle_grads(c, g)
        # the spread between two legal outcomes of the reference itself: its pair sums added in binary32 in one
        # fixed order against the order-free binary64 sums (same terms) -- the floor of any elementwise comparison
        from oracle import oracle as O
        og32 = Hh.oracle_grads(c, O.backward(st, c.gC, c.gD, f32_sums=True))
        ogc = Hh.contracted_oracle_grads(c)
        for k in GRADS:
            ref, got = getattr(og, k), getattr(out.grads, k)
            if ref is None:
                continue
            m, frac = Hh.rel_err(got, ref)
            rec["tensors"]["dL_d" + k] = {"max_rel_8d": float("%.3g" % m), "frac_above_1e-4": float("%.3g" % frac),
                                          "max_err_over_scale": float("%.3g" % Hh.max_err_over_scale(got, ref)),
                                          "elements": int(np.asarray(ref).size)}
            ms, fracs = Hh.rel_err(getattr(out_strict.grads, k), ref)
            rec["tensors"]["dL_d" + k]["strict_gradients"] = {
                "max_rel_8d": float("%.3g" % ms), "frac_above_1e-4": float("%.3g" % fracs),
                "max_err_over_scale": float("%.3g" % Hh.max_err_over_scale(getattr(out_strict.grads, k), ref))}
            m32, frac32 = Hh.rel_err(getattr(og32, k), ref)
            rec["tensors"]["dL_d" + k]["reference_f32_order_vs_f64"] = {
                "max_rel_8d": float("%.3g" % m32), "frac_above_1e-4": float("%.3g" % frac32),
                "max_err_over_scale": float("%.3g" % Hh.max_err_over_scale(getattr(og32, k), ref))}
            if ogc is not None and getattr(ogc, k) is not None:
                # two legal evaluations of the reference's per-pair terms: fp contraction on (nvcc's default) / off
                mc, fracc = Hh.rel_err(getattr(ogc, k), ref)
                rec["tensors"]["dL_d" + k]["reference_contracted_vs_source_order"] = {
                    "max_rel_8d": float("%.3g" % mc), "frac_above_1e-4": float("%.3g" % fracc),
                    "max_err_over_scale": float("%.3g" % Hh.max_err_over_scale(getattr(ogc, k), ref))}
        # side by side, per tensor: share of elements beyond 1e-4 relative in the default mode | in the strict mode | between
        # two legal evaluations of the reference itself (contraction on / off: the floor); and the default's multiple of it
        rec["frac_above_1e-4_default_strict_floor"] = {
            k: [t["frac_above_1e-4"], t["strict_gradients"]["frac_above_1e-4"],
                t.get("reference_contracted_vs_source_order", {}).get("frac_above_1e-4")]
            for k, t in rec["tensors"].items()}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
