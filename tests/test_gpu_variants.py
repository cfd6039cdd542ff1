"""The kernel variants that are selected by environment knobs (read once per process; DESIGN.md, table of tuning
knobs) are held to the same parity tests as the defaults: each one is run in a fresh interpreter on a subset of the
oracle-parity suite that reaches it."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

LARGE_T = ["tests/test_gpu_large_T.py::test_small_n_batched", "tests/test_gpu_large_T.py::test_warm_start_T100"]
SUBSET = ["tests/test_gpu_sym8.py::test_packed_ip1_input", "tests/test_gpu_sym8.py::test_sym8_batched",
          "tests/test_gpu_bench_config.py::test_h2o_shape_t10_against_oracle",
          "tests/test_gpu_bench_config.py::test_k5_every_row_group_body"]


@pytest.mark.parametrize("env", [
    {"EVC_PT_PIPE": "0", "EVC_PT_DMA": "0"}, # phase-alternating pair transform (pt_kernel) instead of ptd_kernel / pt_pipe_kernel
    {"EVC_PT_DMA": "0"},                      # pt_pipe_kernel (operand rows through registers) instead of ptd_kernel (LDS-DMA)
    {"EVC_SUBSPACE_FEW": "0"},                # subspace kernels: no few-roots route (full eigensolver / Jacobi sweeps always)
    {"EVC_LOEWDIN_SPLIT": "0"},               # Loewdin step as one kernel always (no Newton-Schulz X, no riding eigensolver)
    {"EVC_EIGH_F32": "0"},                    # FP64 Jacobi eigensolvers
    {"EVC_EIGH_F32": "1"},                    # FP32 Jacobi start + refinement
    {"EVC_ROWS_LDS": "0"},                    # batched K5 with fragment-shaped loads (gemv_rows_mfma_pipe_kernel)
    # the LDS-staged K5 (gemv_rows_lds_kernel) on the SMALL shapes of the subset (ragged / empty tiles, column tails,
    # images with fewer chunks than slots), in its default shape and in every forced one
    {"EVC_ROWS_LDS_MINCOLS": "1"},
    {"EVC_ROWS_LDS_MINCOLS": "1", "EVC_ROWS_LDS_NT": "14"},
    {"EVC_ROWS_LDS_MINCOLS": "1", "EVC_ROWS_LDS_NT": "7"},
    {"EVC_ROWS_LDS_MINCOLS": "1", "EVC_ROWS_LDS_NT": "4"},
    {"EVC_ROWS_LDS_MINCOLS": "1", "EVC_ROWS_LDS_NT": "2"},
    {"EVC_COLS_LDS": "0"},                    # batched K8 by the row-split / column-tiled kernels of gemv_mfma.hip
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_variant_passes_parity_subset(env):
    e = dict(os.environ)
    e.update(env)
    subset = LARGE_T + SUBSET if "EVC_SUBSPACE_FEW" in env else SUBSET
    if not any(k.startswith(("EVC_ROWS", "EVC_COLS")) for k in env):   # (knobs that do not touch K5 / K8: without its row-group sweep)
        subset = [t for t in subset if "test_k5_every_row_group_body" not in t]
    if "EVC_LOEWDIN_SPLIT" in env:
        subset = subset + ["tests/test_gpu_loewdin_split.py::test_small_batches_take_the_split_form",
                           "tests/test_gpu_warm_start.py::test_warm_start_matches_cold_start[13-5-3-pack2]",
                           "tests/test_gpu_warm_start.py::test_warm_start_matches_cold_start[30-6-30-pack2]"]
    if "EVC_ROWS_LDS_NT" in env or "EVC_ROWS_LDS" in env or "EVC_COLS_LDS" in env:   # (the kernels behind these knobs: batches of >= 12)
        subset = ["tests/test_gpu_bench_config.py::test_k5_every_row_group_body",
                  "tests/test_gpu_bench_config.py::test_k5_row_groups_wide_matrix", "tests/test_gpu_sym8.py::test_sym8_batched"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + subset,
                       cwd=REPO, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
