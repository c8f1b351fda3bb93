"""Reference-CPU <-> port ratio: the reference's own voxelization C++ (oracle/_ref, built
in place by oracle/build_ref.py -- the only part of the path whose reference implementation compiles in this image
without stand-ins) and the oracle's restatement timed on identical inputs in the authoring container.  -> cpu_ratio.json
next to the golden fixtures, so the `cpu_baseline` of bench.py (kind "port") can be read as reference-equivalent time
for these stages.  The sparse-conv stages have no buildable reference here.

    python tests/golden/make_cpu_ratio.py         (needs /root/reference; about a minute)"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def best(fn, n=3):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


def main():
    import oracle
    from isfusion_amd import synthetic
    from oracle import build_ref
    oracle.build()
    ref = build_ref.load_prebuilt() or build_ref.build()
    vs, rg = [0.075, 0.075, 0.2], [-54.0, -54.0, -5.0, 54.0, 54.0, 3.0]
    out = {"host": {"cores": os.cpu_count(), "torch_threads": torch.get_num_threads()}, "cases": []}
    for name, P, T, MV in (("cfg1_20k", 20000, 10, 60000), ("cfg2_frame_300k", 300000, 10, 400000)):
        pts = synthetic.lidar_sweeps(1234 + 1000 + (0 if P == 20000 else 1000), P)
        tp = torch.from_numpy(pts)

        def ref_hard():
            voxels = tp.new_zeros((MV, T, pts.shape[1]))
            coors = tp.new_zeros((MV, 3), dtype=torch.int)
            num = tp.new_zeros((MV,), dtype=torch.int)
            return ref.hard_voxelize(tp, voxels, coors, num, vs, rg, T, MV, 3, True)

        def ref_dyn():
            coors = tp.new_zeros((P, 3), dtype=torch.int)
            ref.dynamic_voxelize(tp, coors, vs, rg, 3)
            return coors

        t_ref_h, t_ref_d = best(ref_hard), best(ref_dyn)
        t_port_h = best(lambda: oracle.hard_voxelize(pts, vs, rg, T, MV))
        t_port_d = best(lambda: oracle.dynamic_voxelize(pts, vs, rg))
        out["cases"].append(dict(case=name, points=P,
                                 hard_voxelize=dict(reference_s=round(t_ref_h, 4), port_s=round(t_port_h, 4),
                                                    port_over_reference=round(t_port_h / t_ref_h, 3)),
                                 dynamic_voxelize=dict(reference_s=round(t_ref_d, 5), port_s=round(t_port_d, 5),
                                                       port_over_reference=round(t_port_d / t_ref_d, 3))))
    json.dump(out, open(os.path.join(HERE, "cpu_ratio.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
