"""Generate tests/golden/encoder_ref.npz: the REFERENCE's SparseEncoder (models/middle_encoders/sparse_encoder.py) built
from the isfusion_0075voxel config kwargs, running through the reference's own Python spconv layer
(ops/bevfusion-ops/spconv/*.py), SparseBasicBlock and make_sparse_convmodule (ops/sparse_block.py) -- with the compiled
module `sparse_conv_ext` (the native boundary this build replaces) served by the CPU oracle, see
ref_harness.install_sparse_encoder.  Pins everything above that boundary: stage wiring, paddings, indice_key sharing,
output-shape rule, residual / BN / ReLU order, dense() layout, state-dict key names.

    python tests/golden/make_golden_encoder.py            # authoring container only
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_harness  # noqa: E402
import isfusion_amd as m  # noqa: E402
import oracle  # noqa: E402
from encoder_common import ENCODER_CASES, encoder_input  # noqa: E402


def main():
    ref = ref_harness.install_sparse_encoder()
    store = {}
    for name, case in ENCODER_CASES.items():
        cfg = dict(case["cfg"])
        mine = m.SparseEncoder(**cfg).eval()
        lb = m.LidarBranch(pts_middle_encoder=cfg)          # only for its seeded weight / BN initialisers
        lb.randomize_weights_(case["seed"]).randomize_bn_(case["seed"] + 1)
        mine.load_state_dict(lb.pts_middle_encoder.state_dict())
        r = ref["sparse_encoder"].SparseEncoder(**cfg).eval()
        missing = r.load_state_dict(mine.state_dict(), strict=True)     # key names and shapes are the reference's
        print(name, "state dict:", len(mine.state_dict()), "tensors loaded into the reference module", missing)
        feats, coors, B = encoder_input(case)
        with torch.no_grad():
            out, enc_feats, _ = r(torch.from_numpy(feats), torch.from_numpy(coors), B)
        obev, outs = oracle.sparse_encoder_forward(mine.plan_to_numpy(), feats, coors, B)
        err = np.abs(out.numpy() - obev).max()
        print(name, tuple(out.shape), "oracle(my plan) vs reference modules:", err, "max", np.abs(obev).max())
        assert out.shape == obev.shape and err < 1e-4
        # per-stage active-voxel counts of the reference's encode_features
        counts = [int(t.features.shape[0]) for t in enc_feats]
        print(name, "voxels per stage", counts)
        nz = np.flatnonzero(obev.reshape(-1))
        g = np.random.default_rng(1)
        pick = np.concatenate([g.choice(nz, min(20000, nz.size), replace=False),
                               g.integers(0, obev.size, 5000)])
        store[f"{name}.idx"] = pick
        store[f"{name}.val"] = out.numpy().reshape(-1)[pick]
        store[f"{name}.shape"] = np.array(out.shape)
        store[f"{name}.stage_voxels"] = np.array(counts)
        store[f"{name}.nonzero"] = np.array([nz.size])
    path = os.path.join(HERE, "encoder_ref.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
