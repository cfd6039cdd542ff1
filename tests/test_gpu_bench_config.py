"""Oracle parity of the EXACT configurations bench.py measures, at full size, plus every row-group body of the
batched K5 kernel.

bench.py's default is H30 (N=30, A=30, T=20), compressed sym8 layout, 32 geometries per batch, int2e / int2e_ip1
handed over packed (aosym s4 / s2kl): 210 rows = 14 row tiles -> the LDS-ring kernel ``gemv_rows_lds_kernel<2,7,2>``
(two row groups of seven tiles), the pair transform ``ptd_kernel<1>`` / ``ptd_kernel<0>`` and ``y2d_kernel``
(csrc/pair_dma.hip); its ``reference_layout`` leg is pack2 at 32 geometries (same K5 kernel over 810 000 columns).  The
kernel names are ASSERTED below from what the library reports having launched (``evc_profile_kernel``), not assumed.
Both are held here to ``oracle.energy_with_grad`` on the ORIGINAL pack2 rows and the full integral arrays
(get_energy_with_grad, ab_initio_gradients_loewdin.py:308-379).  The same for BASELINE configs[4] (Zundel shape:
N=28, AO slices 9,2,2,2,9,2,2, T=30 -> 465 rows) and configs[3] (H2O shape at T=10).

Tolerances: |dE| <= 1e-8 Ha, |dgrad| <= 1e-6 Ha/Bohr (BASELINE.json north_star); the observed differences are
three to four orders of magnitude below them."""
import os

import numpy as np
import pytest
import torch

from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows
from oracle import evcont_oracle as orc

pytestmark = pytest.mark.gpu

E_TOL, G_TOL = 1e-8, 1e-6


def _bundle_from_device(ao):
    c = lambda t: t.cpu().numpy()
    return orc.AOBundle(S=c(ao.S), hcore=c(ao.hcore), eri=c(ao.eri), ipovlp=c(ao.ipovlp), dhcore=c(ao.dhcore),
                        eri_ip1=c(ao.eri_ip1), aoslices=c(ao.aoslices), enuc=ao.enuc, gnuc=c(ao.gnuc))


def _full_size_case(n, A, T, sizes, seed, G, slots, packed_inputs=True):
    """Both legs of bench.py (sym8 with packed integrals, pack2 with full ones) on one batch of G seeded device
    geometries, compared slot by slot with the oracle on the pack2 rows."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator
    from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
    dev = torch.device("cuda:0")
    S, one, rows = make_device_trdm_rows(n, T, 2, seed, dev)
    aos = [make_device_ao(n, A, seed * 1000 + k, dev, sizes, ip1_rs_symmetric=True) for k in range(G)]
    one_h, two_h, S_h = one.cpu().numpy(), rows.cpu().numpy(), S.cpu().numpy()
    want = {k: orc.energy_with_grad(_bundle_from_device(aos[k]), one_h, two_h, S_h) for k in slots}
    del two_h
    got = {}
    # reference layout, full integrals
    trd = DeviceTRDMs.from_device_rows(one, rows, S, 2)
    be = BatchedEvaluator(trd, A, G)
    got["pack2"] = be.energies_with_grads(DeviceAOBatch.stack(aos))
    del be
    # compressed layout, integrals as PySCF's aosym="s4" / "s2kl" would deliver them
    trd.compress_sym8_()
    del rows
    be = BatchedEvaluator(trd, A, G)
    run = [a.packed_ip1(eri=True) for a in aos] if packed_inputs else aos
    got["sym8"] = be.energies_with_grads(DeviceAOBatch.stack(run))
    from evcont_amd import _lib
    ran = {k: _lib.load().evc_profile_kernel(i).decode() for i, k in enumerate(("k5", "k8", "pair_transform"))}
    worst = {"kernels": ran}
    for leg, (E, grad) in got.items():
        de = max(abs(E[k] - want[k][0]) for k in slots)
        dg = max(float(np.abs(grad[k] - want[k][1]).max()) for k in slots)
        worst[leg] = (de, dg)
        assert de <= E_TOL and dg <= G_TOL, (leg, de, dg)
    return worst


def test_h30_bench_default_against_oracle():
    """BASELINE configs[2] = the metric's configuration, exactly as bench.py runs it: G=32, sym8 + packed s4/s2kl
    inputs (K5 gemv_rows_lds_kernel<2,7,2>: row groups 7+7; pair transform ptd_kernel) and pack2."""
    worst = _full_size_case(30, 30, 20, None, 1236, 32, (0, 15, 16, 17, 31))
    ran = worst.pop("kernels")
    if not any(k.startswith("EVC_ROWS_LDS") for k in os.environ):
        assert ran["k5"].startswith("gemv_rows_lds_kernel<2,7,2>"), ran
    if not os.environ.get("EVC_PT_DMA"):
        assert ran["pair_transform"].startswith("ptd_kernel"), ran
    for leg, (de, dg) in worst.items():
        assert de < 1e-10 and dg < 1e-9, (leg, de, dg)     # what the kernels actually deliver


def test_zundel_shape_against_oracle():
    """BASELINE configs[4] shape: N=28, A=7 with AO slices 9,2,2,2,9,2,2, T=30 (465 pair rows = 30 row tiles),
    32 geometries per batch, both layouts."""
    _full_size_case(28, 7, 30, (9, 2, 2, 2, 9, 2, 2), 1238, 32, (0, 13, 31))


def test_h2o_shape_t10_against_oracle():
    """BASELINE configs[3] shape at the T the benchmark assumes: N=13 (slices 9,2,2), T=10, batches of 32 and of 1."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, ContinuationEvaluator
    _full_size_case(13, 3, 10, (9, 2, 2), 1237, 32, (0, 7, 16, 31))
    # MD inner loop: one geometry per call, both layouts
    dev = torch.device("cuda:0")
    S, one, two = make_trdms(13, 10, 77)
    two_p = pack_rows(two, True, True)
    ao = make_ao_arrays(13, 3, 78, ao_sizes=(9, 2, 2), ip1_rs_symmetric=True)
    b = orc.AOBundle(ao.S, ao.hcore, ao.eri, ao.ipovlp, ao.dhcore, ao.eri_ip1, ao.aoslices, ao.enuc, ao.gnuc)
    Eo, go = orc.energy_with_grad(b, one, two_p, S)
    for comp in (None, "sym8"):
        ev = ContinuationEvaluator(DeviceTRDMs(one, two_p, S, dev, compress=comp), 3)
        E, g = ev.energy_with_grad(DeviceAO.from_arrays(ao, dev, pack_ip1=comp is not None, pack_eri=comp is not None))
        assert abs(E - Eo) < 1e-10 and np.abs(g - go).max() < 1e-9, comp


# rows = T(T+1)/2 -> 16-row tiles -> row groups of the batched K5 kernel:
#   two geometry sets, narrow matrix (sym8): groups of <= 7 tiles; wide matrix (pack2, > 200 000 columns) and
#   one geometry set: groups of <= 3 tiles.  T = 12/13/14 -> one group of 5/6/7 tiles, T = 17 -> 5+5,
#   T = 18 -> 6+5, T = 20 -> 7+7; T = 3/7/9 -> one group of 1/2/3 tiles, T = 10 -> 2+2 (ragged last tile).
@pytest.mark.parametrize("T", [3, 7, 9, 10, 12, 13, 14, 15, 17, 18, 20])
@pytest.mark.parametrize("G", [13, 17, 32])
def test_k5_every_row_group_body(T, G):
    """Small N (cheap oracle, 441 columns: below the 4096 the LDS-ring kernel asks for), many training states: every
    ``ntile`` body (1..7) of the fragment-shaped ``gemv_rows_mfma_pipe_kernel`` for one (G=13) and two (G=17, 32)
    geometry sets, the row-split K8 with the same
    row counts, energies AND forces of the first, a middle and the last slot against the oracle."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAO, DeviceAOBatch, BatchedEvaluator
    dev = torch.device("cuda:0")
    n, A = 6, 3
    S, one, two = make_trdms(n, T, 500 + T)
    two_p = pack_rows(two, True, True)
    aos = [make_ao_arrays(n, A, 7000 + 40 * T + k, ip1_rs_symmetric=True) for k in range(G)]
    slots = sorted({0, G // 2, G - 1})
    want = {k: orc.energy_with_grad(orc.AOBundle(aos[k].S, aos[k].hcore, aos[k].eri, aos[k].ipovlp, aos[k].dhcore,
                                                 aos[k].eri_ip1, aos[k].aoslices, aos[k].enuc, aos[k].gnuc),
                                    one, two_p, S) for k in slots}
    for comp in ("sym8", None):
        be = BatchedEvaluator(DeviceTRDMs(one, two_p, S, dev, compress=comp), A, G)
        packed = comp is not None
        E, grad = be.energies_with_grads(DeviceAOBatch.from_arrays(aos, dev, pack_ip1=packed, pack_eri=packed))
        for k in slots:
            assert abs(E[k] - want[k][0]) < 1e-9, (comp, k)
            np.testing.assert_allclose(grad[k], want[k][1], rtol=0, atol=1e-8)


@pytest.mark.parametrize("T,G", [(14, 32), (20, 32), (20, 17), (9, 32)])
def test_k5_row_groups_wide_matrix(T, G):
    """Wide matrices in the reference's own layout: N = 26 in pack2 has 228 826 columns; 105 / 210 / 45 rows = 7 / 14 / 3
    row tiles take the LDS-ring K5 shapes (csrc/gemv_lds.hip lds_pick_nt) and the column-tiled K8 (with EVC_ROWS_LDS=0,
    tests/test_gpu_variants.py, the fragment-shaped kernels with row groups of <= 3 tiles)."""
    from evcont_amd.evaluator import DeviceTRDMs, DeviceAOBatch, BatchedEvaluator
    from evcont_amd.synthetic import make_device_ao, make_device_trdm_rows
    dev = torch.device("cuda:0")
    n, A = 26, 2
    S, one, rows = make_device_trdm_rows(n, T, 2, 900 + T, dev)
    assert rows.shape[1] > 200000
    aos = [make_device_ao(n, A, 9100 + k, dev, ip1_rs_symmetric=True) for k in range(G)]
    one_h, two_h, S_h = one.cpu().numpy(), rows.cpu().numpy(), S.cpu().numpy()
    slots = (0, G - 1)
    want = {k: orc.energy_with_grad(_bundle_from_device(aos[k]), one_h, two_h, S_h) for k in slots}
    be = BatchedEvaluator(DeviceTRDMs.from_device_rows(one, rows, S, 2), A, G)
    E, grad = be.energies_with_grads(DeviceAOBatch.stack(aos))
    for k in slots:
        assert abs(E[k] - want[k][0]) < 1e-9, k
        np.testing.assert_allclose(grad[k], want[k][1], rtol=0, atol=1e-8)
