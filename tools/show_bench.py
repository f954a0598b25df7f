#!/usr/bin/env python3
"""Print the headline and the timed-region diagnostics of bench.py lines (one JSON per file)."""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:   # noqa: BLE001
        print(f, "ERR", e); continue
    c = d.get("config", {})
    print(f"{f}: value {d.get('value')} ms/step {d.get('ms_per_step')} median {c.get('ms_per_step_median', d.get('ms_per_step_median'))} "
          f"first {c.get('step_ms_first')} max {c.get('step_ms_max')} over1.15x {c.get('steps_over_1p15x_median')} "
          f"host max/med {c.get('host_max_ms_per_step')}/{c.get('host_median_ms_per_step')} tail {c.get('closing_fence_tail_ms')} "
          f"sum_stage {c.get('sum_stage_ms')} bwd {d.get('roofline', {}).get('launch_ms')}")
    sec = d.get("secondary") or {}
    print("   secondary:", {k: v.get("ms_per_step") for k, v in sec.items()})
