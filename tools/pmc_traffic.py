#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a pmc_summary.json (tools/pmc_summary.py): measured HBM bytes per
launch of every bench stage = sum over the stage's kernels of (2 x FETCH_SIZE + WRITE_SIZE) KiB,
FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc passes and FETCH_SIZE doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950.  bench.py reads the result for
`roofline.traffic`.

    python tools/pmc_traffic.py profiles/<round>/pmc_summary.json [config] > profiles/pmc_traffic.json

Run it in the same tree the passes were taken on: the output records `csrc_sha256`, the hash of the kernel sources
(bench.csrc_sha256), and bench.py reports `roofline.traffic` only while that hash matches the library it is timing.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STAGE_OF = {   # kernel name prefix (template arguments stripped) -> bench stage
    "bsr::k_preprocess_bwd": "preprocess_bwd", "bsr::k_preprocess": "preprocess", "bsr::k_scans": "scan_wg", "bsr::k_emit": "binning", "bsr::k_radix_hist": "binning", "bsr::k_radix_rowscan": "binning",
    "bsr::k_radix_scatter": "binning", "bsr::k_tile_ranges": "binning", "bsr::k_tile_count": "binning",
    "bsr::k_tile_starts": "binning", "bsr::k_tile_scatter": "binning", "bsr::k_sort_tiles": "sort_tiles",
    "bsr::k_bucket_sort": "sort_tiles",   # (second binning pass + per-tile sort in one launch: timed as the sort stage)
    "bsr::k_render_fwd": "render_fwd", "bsr::k_render_bwd": "render_bwd",
}
# kernels launched more than once per step: launches per step (radix passes at 1080p after the one fused
# with the emit: 1)
PER_STEP = {"bsr::k_radix_hist": 1, "bsr::k_radix_rowscan": 1, "bsr::k_radix_scatter": 1}
SIMDS = 1024   # 256 CUs x 4


def main():
    src = sys.argv[1]
    config = sys.argv[2] if len(sys.argv) > 2 else "c3"
    summ = json.load(open(src))
    stages = {}
    for kern, r in summ.items():
        base = kern.split("<")[0]
        stage = next((s for p, s in sorted(STAGE_OF.items(), key=lambda kv: -len(kv[0])) if base.startswith(p)), None)
        if stage is None:
            continue
        mult = PER_STEP.get(base, 1)
        rd = 2.0 * r.get("FETCH_SIZE", 0.0) * 1024.0 * mult
        wr = r.get("WRITE_SIZE", 0.0) * 1024.0 * mult
        st = stages.setdefault(stage, {"hbm_bytes": 0, "read_bytes_2xFETCH_SIZE": 0, "write_bytes": 0,
                                       "profiled_us": 0.0})
        if stage in ("render_fwd", "render_bwd") and r.get("GRBM_GUI_ACTIVE") and r.get("SQ_INSTS_VALU"):
            # one kernel per stage; GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles -> core clock during the kernel
            cycles = r["GRBM_GUI_ACTIVE"] / 8.0
            st["core_clock_mhz"] = round(cycles / r["dur_us(profiled)"])
            st["valu_insts_per_simd_cycle"] = round(r["SQ_INSTS_VALU"] / (SIMDS * cycles), 4)
        st["hbm_bytes"] += int(rd + wr)
        st["read_bytes_2xFETCH_SIZE"] += int(rd)
        st["write_bytes"] += int(wr)
        st["profiled_us"] = round(st["profiled_us"] + r.get("dur_us(profiled)", 0.0) * mult, 1)
    from bench import csrc_sha256
    out = {config: stages,
           "profile": os.path.basename(os.path.dirname(os.path.abspath(src))),
           "csrc_sha256": csrc_sha256(),
           "_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 3 "
                    "--warmup 1` (config " + config + ") on MI355X; bytes per launch = mean over launches, summed "
                    "over the kernels of a stage; read bytes = 2 x FETCH_SIZE KiB (gfx950 reports half of wide "
                    "coalesced reads, MI355X_MICROARCH.md section HBM) -- an upper bound for kernels whose reads are "
                    "narrow gathers (the tile renderers); Infinity-Cache hits are included in FETCH_SIZE; "
                    "valu_insts_per_simd_cycle = SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): the tile renderers "
                    "issue one VALU instruction per ~3.5 cycles and SIMD, which with the measured issue costs of their "
                    "instruction mix (tools/microbench/valu_rates.hip: 2.6 .. 4.25 cycles) is a saturated VALU port; source: "
                    + src}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
