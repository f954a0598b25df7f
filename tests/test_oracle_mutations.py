"""Mutation check of the (reference-unpinned) CPU oracle: does the test-suite have teeth?

The reference holds no golden vectors for this path and cannot be built here, so oracle/bsr_oracle.c is pinned only
by the float64-autograd cross-check (test_oracle_crosscheck.py, an independent restatement written from the maths)
and by the hand-derived closed-form micro-cases (test_oracle_micro.py).  This test breaks the oracle the way a
transcription slip would -- one token per mutant: a constant of SURVEY.md row A17, a sign, an index, a factor --
builds each broken copy and requires that cross-check + micro-cases FAIL on it.  A mutant nobody notices is a
line of the oracle that nothing pins.

    python tests/test_oracle_mutations.py --report profiles/r02_v3/oracle_mutation_report.json     (full table)

Mutants that are expected to survive are listed with the reason (their effect is below fp32 resolution).
"""
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "oracle", "bsr_oracle.c")
CFLAGS = ["-O2", "-fPIC", "-std=c99", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared"]
DETECTORS = ["tests/test_oracle_micro.py", "tests/test_oracle_crosscheck.py"]

# (name, reference lines pinned, old text, new text, which occurrence (0-based))
MUTANTS = [
    ("lowpass_0.3_x", "forward.cu:110", "c.c[0][0] += 0.3f;", "c.c[0][0] += 0.2f;", 0),
    ("lowpass_0.3_y", "forward.cu:111", "c.c[1][1] += 0.3f;", "c.c[1][1] += 0.2f;", 0),
    ("lowpass_bwd_0.3", "backward.cu:197-199", "float a = cov2D.c[0][0] += 0.3f;", "float a = cov2D.c[0][0] += 0.2f;", 0),
    ("guard_band_1.3", "forward.cu:82-87", "const float limx = 1.3f * tan_fovx;", "const float limx = 1.0f * tan_fovx;", 0),
    ("x_grad_mul", "backward.cu:175", "*xmul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;", "*xmul = (txtz < -limx || txtz > limx) ? 1.0f : 1.0f;", 0),
    ("y_grad_mul_sign", "backward.cu:176", "*ymul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;", "*ymul = (tytz < -limy || tytz > limy) ? 1.0f : 0.0f;", 0),
    ("near_plane_0.2", "auxiliary.h:154", "if (p_view->z <= 0.2f)", "if (p_view->z <= 0.3f)", 0),
    ("radius_3_sigma", "forward.cu:232", "ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)))", "ceilf(2.f * sqrtf(fmaxf(lambda1, lambda2)))", 0),
    ("lambda_floor_0.1", "forward.cu:230", "float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));", "float lambda1 = mid + sqrtf(fmaxf(1.1f, mid * mid - det));", 0),
    ("conic_index_order", "forward.cu:223", "{cov[2] * det_inv, -cov[1] * det_inv, cov[0] * det_inv}", "{cov[0] * det_inv, -cov[1] * det_inv, cov[2] * det_inv}", 0),
    ("conic_b_sign", "forward.cu:223", "-cov[1] * det_inv", "cov[1] * det_inv", 0),
    ("ndc2pix_minus_1", "auxiliary.h:43", "(((v + 1.0) * S - 1.0) * 0.5)", "(((v + 1.0) * S - 0.0) * 0.5)", 0),
    ("rect_block_minus_1", "auxiliary.h:52", "(px + max_radius + BLOCK_X - 1) / BLOCK_X", "(px + max_radius + BLOCK_X) / BLOCK_X", 0),
    ("proj_matrix_index", "auxiliary.h:72", "o[0] = m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12];", "o[0] = m[0] * p.x + m[5] * p.y + m[8] * p.z + m[12];", 0),
    ("sh_c0", "auxiliary.h:22", "SH_C0 = 0.28209479177387814f", "SH_C0 = 0.38209479177387814f", 0),
    ("sh_deg1_sign", "forward.cu:44", "r = r - SH_C1 * y * SH(1) + SH_C1 * z * SH(2) - SH_C1 * x * SH(3);", "r = r - SH_C1 * y * SH(1) - SH_C1 * z * SH(2) - SH_C1 * x * SH(3);", 0),
    ("sh_c2_sign", "auxiliary.h:26", "{1.0925484305920792f, -1.0925484305920792f,", "{1.0925484305920792f, 1.0925484305920792f,", 0),
    ("sh_c3_constant", "auxiliary.h:34", "0.3731763325901154f", "0.4731763325901154f", 0),
    ("sh_offset_0.5", "forward.cu:63", "r += 0.5f;", "r += 0.4f;", 0),
    ("sh_bwd_deg1_x", "backward.cu:60", "dRGBdx[ch] = -SH_C1 * SH(3);", "dRGBdx[ch] = SH_C1 * SH(3);", 0),
    ("sh_bwd_deg3_factor", "backward.cu:113", "SH_C3[0] * SH(9) * 3.f * 2.f * xy", "SH_C3[0] * SH(9) * 3.f * 1.f * xy", 0),
    ("dnormvdv_sign", "auxiliary.h:113", "d.x = ((+sum2 - v.x * v.x) * dv.x - v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;", "d.x = ((+sum2 - v.x * v.x) * dv.x + v.y * v.x * dv.y - v.z * v.x * dv.z) * invsum32;", 0),
    ("alpha_clamp_0.99", "forward.cu:426", "float alpha = fminf(0.99f, con_o[3] * bsro_expf(power));", "float alpha = fminf(0.98f, con_o[3] * bsro_expf(power));", 0),
    ("alpha_threshold_255", "forward.cu:427", "if (alpha < 1.0f / 255.0f) continue;\n\t\t\t\t\tfloat test_T", "if (alpha < 1.0f / 256.0f) continue;\n\t\t\t\t\tfloat test_T", 0),
    ("T_stop_1e-4", "forward.cu:434", "if (test_T < 0.0001f) break;", "if (test_T < 0.001f) break;", 0),
    ("power_half", "forward.cu:419", "float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - con_o[1] * dx * dy;", "float power = -0.5f * (con_o[0] * dx * dx + con_o[2] * dy * dy) - 0.5f * con_o[1] * dx * dy;", 0),
    ("depth_gate_0.5", "forward.cu:464", "(acc > 0.5f) ? Dd / acc : 0", "(acc > 0.25f) ? Dd / acc : 0", 0),
    ("depth_acc_seed", "forward.cu:387", "float acc = 0.000001f;", "float acc = 0.0f;", 0),
    ("bwd_clamp_0.99", "backward.cu:513", "const float alpha = fminf(0.99f, con_o[3] * G);\n\t\t\t\t\tif (alpha < 1.0f / 255.0f) continue;\n\t\t\t\t\tT = T / (1.f - alpha);\n\t\t\t\t\tconst float dchannel_dcolor", "const float alpha = fminf(0.98f, con_o[3] * G);\n\t\t\t\t\tif (alpha < 1.0f / 255.0f) continue;\n\t\t\t\t\tT = T / (1.f - alpha);\n\t\t\t\t\tconst float dchannel_dcolor", 0),
    ("bwd_bg_term_sign", "backward.cu:557", "dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot_dpixel;", "dL_dalpha += (T_final / (1.f - alpha)) * bg_dot_dpixel;", 0),
    ("bwd_accum_rec", "backward.cu:529", "accum_rec[ch] = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];", "accum_rec[ch] = last_alpha * accum_rec[ch] + (1.f - last_alpha) * last_color[ch];", 0),
    ("ddelx_dx", "backward.cu:473", "const float ddelx_dx = 0.5 * W; /* :473 */", "const float ddelx_dx = 0.5 * H; /* :473 */", 0),
    ("dconic_xy_half", "backward.cu:579", "const float v3 = -0.5f * gdx * dy * dL_dG;", "const float v3 = -1.0f * gdx * dy * dL_dG;", 0),
    ("dG_ddely_index", "backward.cu:564", "const float dG_ddely = -gdy * con_o[2] - gdx * con_o[1];\n\t\t\t\t\tconst float v0", "const float dG_ddely = -gdy * con_o[0] - gdx * con_o[1];\n\t\t\t\t\tconst float v0", 0),
    ("denom2inv_eps", "backward.cu:203", "float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);", "float denom2inv = 1.0f / ((denom * denom) + 0.0f);", 0),
    ("dL_db_factor_2", "backward.cu:212", "dL_db = denom2inv * 2 * (b * c * dL_dconic[0]", "dL_db = denom2inv * 1 * (b * c * dL_dconic[0]", 0),
    ("dcov_offdiag_2", "backward.cu:224", "dL_dcov[6 * idx + 1] = 2 * Tm(0,0) * Tm(0,1) * dL_da", "dL_dcov[6 * idx + 1] = 1 * Tm(0,0) * Tm(0,1) * dL_da", 0),
    ("dL_dtz_2hx", "backward.cu:264", "(2 * h_x * t.x) * tz3 * dL_dJ02", "(1 * h_x * t.x) * tz3 * dL_dJ02", 0),
    ("dmean_transpose", "backward.cu:268", "vec3 dL_dmean = transformVec4x3Transpose(d, view_matrix);", "vec3 dL_dmean = transformPoint4x3(d, view_matrix);", 0),
    ("dSigma_offdiag_half", "backward.cu:308-312", "mat3 dL_dSigma = m3(d[0], 0.5f * d[1],", "mat3 dL_dSigma = m3(d[0], 1.0f * d[1],", 0),
    ("dscale_with_mod", "backward.cu:322-325", "dL_dscale[0] = col_dot(Rt.c[0], dL_dMt.c[0]);", "dL_dscale[0] = mod * col_dot(Rt.c[0], dL_dMt.c[0]);", 0),
    ("dMt_scale_index", "backward.cu:328", "dL_dMt.c[1][k] *= s[1];", "dL_dMt.c[1][k] *= s[0];", 0),
    ("dquat_factor_4", "backward.cu:334", "- 4 * x * (G(2,2) + G(1,1));", "- 2 * x * (G(2,2) + G(1,1));", 0),
    ("proj_bwd_mul2", "backward.cu:376", "float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w * m_w;", "float mul2 = (proj[1] * m.x + proj[5] * m.y + proj[9] * m.z + proj[13]) * m_w;", 0),
    ("quat_not_normalised_xy", "forward.cu:134", "2.f * (x * y - r * z)", "2.f * (x * y + r * z)", 0),
    # expected survivors: the effect is below what fp32 arithmetic resolves on any input of the path
    ("eps_p_w_1e-7", "forward.cu:199", "float p_w = 1.0f / (p_hom[3] + 0.0000001f);", "float p_w = 1.0f / (p_hom[3] + 0.0f);", 0),
    ("eps_m_w_1e-7", "backward.cu:374", "float m_w = 1.0f / (m_hom[3] + 0.0000001f);", "float m_w = 1.0f / (m_hom[3] + 0.0f);", 0),
]
EXPECTED_SURVIVORS = {
    "eps_p_w_1e-7": "w = view depth > 0.2, so 1e-7 is <= 5e-7 relative (a few ulp) on the near plane and below one ulp "
                    "from z = 2 on; no output moves by more than fp32 rounding noise",
    "eps_m_w_1e-7": "same quantity in the backward (m_w = 1 / (w + 1e-7), w > 0.2)",
}


def _mutate(text, old, new, which):
    pos, start = -1, 0
    for _ in range(which + 1):
        pos = text.find(old, start)
        if pos < 0:
            return None
        start = pos + 1
    return text[:pos] + new + text[pos + len(old):]


def run_mutant(m, tmp, src_text):
    name, cites, old, new, which = m
    mutated = _mutate(src_text, old, new, which)
    if mutated is None:
        return name, "missing", "source text not found: the mutation list is stale"
    c = os.path.join(tmp, name + ".c")
    so = os.path.join(tmp, name + ".so")
    with open(c, "w") as fh:
        fh.write(mutated)
    subprocess.check_call(["gcc"] + CFLAGS + ["-o", so, c, "-lm"])
    env = dict(os.environ, BSR_ORACLE_LIB=so, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + DETECTORS,
                       cwd=ROOT, env=env, capture_output=True, text=True)
    first = [ln for ln in r.stdout.splitlines() if ln.startswith("FAILED")]
    return name, ("killed" if r.returncode != 0 else "survived"), (first[0][:160] if first else "")


def run_all(workers=4):
    src_text = open(SRC).read()
    with tempfile.TemporaryDirectory() as tmp:
        # the unmutated source through the same pipeline must pass, or a "kill" means nothing
        base = run_mutant(("unmutated", "", "static const float SH_C0", "static const float SH_C0", 0), tmp, src_text)
        with ThreadPoolExecutor(workers) as ex:
            res = list(ex.map(lambda m: run_mutant(m, tmp, src_text), MUTANTS))
    return base, res


def test_every_mutant_of_the_oracle_is_caught():
    base, res = run_all()
    assert base[1] == "survived", f"the unmutated oracle fails its own detectors: {base}"
    bad = [r for r in res if r[1] == "missing"]
    assert not bad, bad
    survivors = sorted(r[0] for r in res if r[1] == "survived")
    assert survivors == sorted(EXPECTED_SURVIVORS), f"unexpected mutation survivors / kills: {survivors}"
    assert len(res) - len(survivors) >= 40


if __name__ == "__main__":
    base, res = run_all(workers=int(os.environ.get("WORKERS", "6")))
    cites = {m[0]: m[1] for m in MUTANTS}
    table = [{"mutant": n, "reference_lines": cites[n], "result": st, "first_failing_check": why,
              **({"why_it_survives": EXPECTED_SURVIVORS[n]} if n in EXPECTED_SURVIVORS else {})} for n, st, why in res]
    killed = sum(1 for r in res if r[1] == "killed")
    out = {"unmutated_passes": base[1] == "survived", "mutants": len(res), "killed": killed,
           "score": round(killed / len(res), 4), "detectors": DETECTORS, "table": table}
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        os.makedirs(os.path.dirname(sys.argv[2]), exist_ok=True)
        with open(sys.argv[2], "w") as fh:
            json.dump(out, fh, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "table"}))
    for r in table:
        print("%-26s %-9s %s" % (r["mutant"], r["result"], r["first_failing_check"]))
