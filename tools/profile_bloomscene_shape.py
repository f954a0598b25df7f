#!/usr/bin/env python3
"""The BloomScene-shaped secondary workload of bench.py on its own (100 k anchors x 10 offsets through the fused anchor
expansion into the rasterizer with colors_precomp, 512 x 512, fwd+bwd), for rocprofv3 --kernel-trace --stats:

    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/profile_bloomscene_shape.py
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="default", choices=["default", "capacity", "graph"],
                    help="default: the reference's call shape; capacity: static shapes, no host wait; graph: that step "
                         "replayed from a HIP graph (bench.bloomscene_shape_workload)")
    a = ap.parse_args()
    args = argparse.Namespace(steps=a.steps, warmup=a.warmup, prewarm_ms=300.0)
    D = bench.Dist(1)
    print(json.dumps(bench.bloomscene_shape_workload(D, args, mode=a.mode)))
    D.close()


if __name__ == "__main__":
    main()
