"""Child process of tests/test_gpu_lds_kernels.py: batched energies and forces of a list of shapes with the streaming
kernels the environment selects (EVC_ROWS_LDS / EVC_COLS_LDS: the LDS-staged K5 / K8 of csrc/gemv_lds.hip, or the
fragment-shaped kernels of csrc/gemv_mfma.hip), written to an .npz.  The inputs are generated on the device from seeds,
so two children see identical data."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch

# (N, A, T, G, layout): N = 30 / T = 20 is the benchmark shape; T = 23 -> 276 rows (18 tiles: row groups of 4 + padding),
# T = 5 -> one ragged tile; G = 44 = one pass of 32 + one of 12; pack2 = the reference's layout (wide matrix)
SHAPES = [(30, 4, 20, 32, "sym8"), (30, 4, 20, 17, "sym8"), (30, 3, 23, 29, "sym8"), (24, 3, 12, 44, "sym8"),
          (32, 3, 5, 13, "sym8"), (20, 2, 14, 32, "sym8"), (26, 2, 9, 32, "pack2"), (17, 3, 16, 31, "sym8"),
          # more than 32 geometries: K5 takes up to 64 per pass (four geometry sets), K8 passes of 32
          (30, 3, 20, 64, "sym8"), (30, 2, 20, 50, "sym8"), (22, 2, 23, 45, "sym8"), (16, 2, 7, 76, "sym8"),
          # tall matrices (more rows than weights fit in LDS): K8 in row slabs with LDS-DMA-staged weights
          (16, 2, 40, 32, "sym8"), (18, 2, 48, 17, "sym8")]


def main(out_path):
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator
    from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
    dev = torch.device("cuda:0")
    res = {}
    for k, (n, A, T, G, lay) in enumerate(SHAPES):
        S, one, rows = make_device_trdm_rows(n, T, 2, 3100 + k, dev)
        trd = DeviceTRDMs.from_device_rows(one, rows, S, 2)
        del rows
        packed = lay == "sym8"
        if packed:
            trd.compress_sym8_()
        aos = [make_device_ao(n, A, 5200 + 50 * k + j, dev, ip1_rs_symmetric=True) for j in range(G)]
        if packed:
            aos = [a.packed_ip1(eri=True) for a in aos]
        be = BatchedEvaluator(trd, A, G)
        E, grad = be.energies_with_grads(DeviceAOBatch.stack(aos))
        res[f"E{k}"] = np.asarray(E)
        res[f"g{k}"] = np.asarray(grad)
        del be, trd, aos
        torch.cuda.empty_cache()
    np.savez(out_path, **res)


if __name__ == "__main__":
    main(sys.argv[1])
