"""bench.py on the GPU box: the JSON line of the default invocation carries the contract keys, the roofline record,
the strong-scaled C4 sweep and the secondary workloads; and the N > 1 code path (RCCL broadcast, barriers, gradient
all-reduce) executes with one rank under torch.distributed.run."""
import os

import pytest

pytestmark = pytest.mark.gpu

MAX_LINE_BYTES = 4096   # the driver reads the LAST stdout line; round 5's 22 KB line left BENCH_r05.parsed = null


def _run_bench(extra, env_extra=None, nproc=None, want_line=False):
    """bench.py as a child process (optionally under torch.distributed.run with one rank).  Checks the contract of
    its stdout -- the LAST line is one compact JSON object -- and returns the DETAIL record it wrote (a superset of the
    line: per-step arrays, the whole C4 record, the secondary legs); want_line: (line, detail)."""
    import json
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **(env_extra or {}))
    cmd = [sys.executable]
    if nproc:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                "--master-port", "29533"]
    with tempfile.TemporaryDirectory() as tmp:
        detail_path = os.path.join(tmp, "detail.json")
        cmd += [os.path.join(root, "bench.py")] + extra + ["--detail", detail_path]
        r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.strip(), (r.returncode, r.stdout[-500:], r.stderr[-1500:])
        last = r.stdout.rstrip("\n").splitlines()[-1]
        assert len(last.encode()) < MAX_LINE_BYTES, len(last)
        line = json.loads(last)                       # the last line alone must parse
        assert line["detail"] == detail_path
        with open(detail_path) as fh:
            detail = json.load(fh)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "ms_per_step", "config", "roofline", "stage_ms"):
        assert line[k] == detail[k] or k in ("config", "roofline"), k
    return (line, detail) if want_line else detail


def test_bench_rccl_path_on_one_rank():
    """bench.py's N > 1 code path -- RCCL init, the packed broadcast of the Gaussian buffers, barriers, the packed
    gradient all-reduce of the data-parallel step and the strong-scaled C4 sweep -- run with ONE
    rank under torch.distributed.run (BSR_BENCH_FORCE_DIST=1), on a reduced Gaussian count.  8-GPU runs are the driver's;
    this proves the collectives execute on RCCL and the JSON carries the C4 record."""
    d = _run_bench(["--gpus", "1", "--steps", "6", "--warmup", "2", "--gaussians", "200000", "--allreduce-grads",
                    "--no-cpu-baseline"], env_extra={"BSR_BENCH_FORCE_DIST": "1"}, nproc=1)
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["config"]["broadcast_ms"] > 0                       # the broadcast really ran
    assert d["config"]["allreduce_ms_per_step"] > 0              # and so did the all-reduce, every step
    assert "gradient all-reduce" in d["config"]["parallelism"]
    c4 = d["c4"]
    assert c4["scaling"] == "strong" and c4["views"] == 64 and c4["views_per_rank"] == [64]
    assert c4["broadcast_ms"] > 0 and c4["broadcast_bytes"] == 1_000_000 * (3 + 3 + 4 + 1 + 48) * 4
    for k in ("views_per_call_1", "views_per_call_16"):
        assert c4[k]["value"] > c4[k]["value_including_broadcast"] > 0
    assert "secondary" not in d                                  # not the headline workload


@pytest.mark.parametrize("world", [2, 4])
def test_bench_several_ranks_control_flow_on_one_gpu(world):
    """The N > 1 control flow of bench.py -- scene generated on rank 0 only, packed broadcast into the other ranks' empty
    buffers, per-rank cameras, views dealt round-robin, barrier + max-over-ranks timing, rank 0 printing the compact line
    with world_size and every rank's device -- with all ranks on the one GPU of this box over gloo (RCCL refuses two
    ranks on one device; the RCCL calls themselves are exercised by test_bench_rccl_path_on_one_rank).  Four ranks is
    what a one-GPU box allows comfortably (at most six processes may hold the card); the driver's N = 8 run differs
    only in the number of ranks.  A reduced Gaussian count keeps it short.  With two ranks also the experimental
    distribution forms (--experimental --c4-forms all)."""
    extra = ["--experimental", "--c4-forms", "all"] if world == 2 else []
    line, d = _run_bench(["--gpus", str(world), "--steps", "6", "--warmup", "2", "--gaussians", "200000", "--allreduce-grads",
                          "--no-cpu-baseline"] + extra,
                         env_extra={"BSR_BENCH_SINGLE_DEVICE": "1", "BSR_BENCH_BACKEND": "gloo"}, nproc=world, want_line=True)
    sv = d["c4"]["scatter_visible"]
    if world == 2:
        assert sv["distribution_ms_pipelined"] > 0 and sv["distribution_ms_blockwise"] > 0 and "predicted" in d["c4"]
    else:
        assert "distribution_ms_pipelined" not in sv and "predicted" not in d["c4"]
    assert d["n_gpus"] == world and d["scaling"] == "weak"
    assert line["world_size"] == world and line["devices"] == [0] * world
    assert d["value"] > 0 and d["config"]["visible"] > 150_000          # rank 0 rendered the broadcast scene
    assert d["config"]["broadcast_ms"] > 0 and d["config"]["allreduce_ms_per_step"] > 0
    c4 = d["c4"]
    assert c4["n_gpus"] == world and c4["views_per_rank"] == [64 // world] * world and c4["broadcast_ms"] > 0
    assert c4["views_per_call_16"]["value_including_broadcast"] > 0
    s4 = line["c4"]
    assert s4["views_per_rank"] == [64 // world] * world and s4["sweep_ms_cold"] > s4["sweep_ms_resident"] > 0
    assert s4["scatter_visible"]["distribution_ms"] > 0
    assert "secondary" not in d and "cpu_baseline" not in d


def test_bench_launches_its_own_ranks_when_typed_plainly():
    """`python bench.py --gpus 2 ...` WITHOUT a launcher (the form the driver uses for N = 1): the parent must start the
    two ranks itself as child processes under torch.distributed.run and relay rank 0's JSON line (round 2: it exited
    with rc 1).  Both ranks on this box's one GPU over gloo, as in the control-flow test above."""
    d = _run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1", "--gaussians", "100000", "--no-cpu-baseline",
                    "--prewarm-ms", "20"], env_extra={"BSR_BENCH_SINGLE_DEVICE": "1", "BSR_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and d["steps"] == 4
    assert d["c4"]["n_gpus"] == 2 and d["c4"]["views_per_rank"] == [32, 32]


def test_bench_headline_line_has_the_contract_keys():
    """The default invocation (shortened): the LAST stdout line is a compact (< 4 KB) JSON object with the contract
    keys, config.workload, roofline and cpu_baseline -- what the driver parses into BENCH_rNN.json -- and the detail
    record beside it holds the C4 sweep and the secondary workloads."""
    line, d = _run_bench(["--steps", "8", "--warmup", "2", "--cpu-sample", "20000"], want_line=True)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "roofline_step", "stage_ms", "cpu_baseline"):
        assert k in line, k
    assert line["metric"].startswith("Msplats/s fwd+bwd @1M Gaussians 1920x1080 SH3")
    assert line["steps"] == 8 and line["warmup"] == 2 and line["n_gpus"] == 1 and line["dtype"] == "f32"
    assert line["vs_baseline"] is None and line["higher_is_better"] is True and line["scaling"] == "weak"
    cfg = line["config"]
    assert cfg["workload"].startswith("c3: 1000000 Gaussians, SH deg 3, 1920x1080, fwd+bwd")
    assert cfg["gaussians"] == 1_000_000 and cfg["num_rendered"] > 4_000_000 and "model" not in cfg
    assert len(cfg["csrc_sha256"]) == 16 and cfg["exact_exp"] == 0 and cfg["strict_gradients"] == 0
    for k in ("ms_per_step_median", "step_ms_first", "step_ms_max", "host_max_ms_per_step"):
        assert cfg[k] > 0, k
    assert abs(line["value"] - 1e-3 * cfg["gaussians"] / line["ms_per_step"]) < 0.01 * line["value"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "render_bwd" and 0 < rf["frac"] < 1 and rf["peak"] == 8000.0
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-4
    assert abs(rf["algorithmic_bytes"] / (rf["launch_ms"] * 1e-3) / 1e9 - rf["achieved"]) < 0.01 * rf["achieved"]
    # inside the timed region only the dominant stage carries events (every 4th of the 8 steps); the stage table is complete
    assert rf["launches_timed"] == 2 and d["roofline"]["timed_region_events"]["stage"] == "render_bwd"
    assert set(line["stage_ms"]) >= {"preprocess", "scan_wg", "binning", "sort_tiles", "render_fwd", "render_bwd",
                                     "preprocess_bwd"}
    # the committed counter passes count only while they belong to the kernel sources being timed
    src = d["roofline"]["traffic_source"]
    assert src is None or (rf["traffic"] is not None) == src["matches_timed_build"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Msplats/s" and cb["sample"]
    # the C4 summary of the line, and the whole record in the detail file
    c4 = line["c4"]
    assert c4["views"] == 64 and c4["views_per_rank"] == [64] and c4["broadcast_ms"] == 0.0
    assert c4["value_resident"] > 0 and c4["sweep_ms_cold"] >= c4["sweep_ms_resident"] > 0
    assert d["c4"]["views_per_rank"] == [64] and "predicted" not in d["c4"]        # (the model: --experimental only)
    sec = d["secondary"]
    assert sec["c3_dense_scales_x3"]["instances_per_gaussian"] > 10       # long tile lists
    assert sec["c3_camera_changes_every_step"]["value"] > 0 and sec["bloomscene_shape"]["value"] > 0
    # the strict-gradient and exact-exp modes are on the record, beside the default they are slower than
    assert 0 < sec["c3_strict_gradients"]["value"] and 0 < sec["c3_exact_exp"]["value"]
    assert line["secondary_Msplats_per_s"]["c3_strict_gradients"] == sec["c3_strict_gradients"]["value"]
    assert d["config"]["device_allocs_in_timed_region"] == 0
    assert len(d["step_ms"]) == 8


def test_profile_only_brackets_one_stage():
    """bsr_profile_only(stage): the library records events for that stage alone (what bench.py relies on to time the
    dominant kernel inside the timed region without taxing the step), and NULL restores all stages."""
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    import helpers as Hh
    from bloomscene_amd import _capi
    c = Hh.make_case(P=20_000, W=256, H=192, deg=1, seed=5)
    try:
        _capi.profile_only("render_fwd")
        _capi.profile_enable(1)
        _capi.profile_reset()
        Hh.run_hip(c)
        Hh.run_hip(c)
        torch.cuda.synchronize()
        prof = {k: v for k, v in _capi.profile_read().items() if v[1] > 0}
        assert set(prof) == {"render_fwd"} and prof["render_fwd"][1] == 2 and prof["render_fwd"][0] > 0
        _capi.profile_only(None)
        _capi.profile_reset()
        Hh.run_hip(c)
        torch.cuda.synchronize()
        prof = {k: v for k, v in _capi.profile_read().items() if v[1] > 0}
        assert {"preprocess", "scan_wg", "binning", "sort_tiles", "render_fwd"} <= set(prof)
    finally:
        _capi.profile_enable(False)
        _capi.profile_only(None)
