"""The PySCF-facing side of the path, executed against a PySCF-shaped stand-in
(tests/pyscf_stub.py; this image has no PySCF): ``integrals.ao_arrays(Mole)``, ``get_energy_with_grad(Mole, ...)``
against the reference's golden outputs, ``get_scanner`` as a ``lib.GradScanner`` driven by an integrator with PySCF's
control flow (one ``scanner(mol)`` per step after ``mol.set_geom_``; ``callback(locals())`` reading
``locals["scanner"].base.predicted_one_rdm``, 04_Zundel_continuation_MD.py:140-177), ``get_trajectory`` through the
``pyscf.md.NVE`` branch (MD_utils.py:99-120), and the three example drivers at reduced size."""
import os
import subprocess
import sys

import numpy as np
import pytest

import pyscf_stub
from conftest import ao_from_golden, golden_cases

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case", golden_cases())
def test_mole_queries_and_energy_with_grad_golden(case, load_golden):
    """A Mole-shaped object serving the golden AO arrays: the adapter makes exactly the queries the reference makes
    and the result is the reference's own output for those arrays."""
    import evcont_amd.ab_initio_gradients_loewdin as gl
    from evcont_amd.integrals import ao_arrays, is_array_mol
    g = load_golden(case)
    with pyscf_stub.installed():
        mol = pyscf_stub.StubMole(ao_from_golden(g))
        assert not is_array_mol(mol)
        ao = ao_arrays(mol, need_grad=True)
        for f in ("S", "hcore", "eri", "ipovlp", "dhcore", "eri_ip1", "gnuc"):
            assert np.array_equal(np.asarray(getattr(ao, f)).reshape(-1), np.asarray(g[f]).reshape(-1)), f
        assert np.array_equal(ao.aoslices, g["aoslices"]) and ao.enuc == float(g["enuc"])
        assert {q[0] for q in mol.queries} == {"int1e_ovlp", "int2e", "int1e_ipovlp", "int2e_ip1"}
        E, grad, D, G = gl.get_energy_with_grad(mol, g["one_RDM"], g["two_RDM"], g["S_train"],
                                                return_density_matrices=True)
    assert abs(E - float(g["ewg_E_full6"])) < 1e-10
    np.testing.assert_allclose(grad, g["ewg_grad_full6"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(D, g["ewg_D_full6"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(G, g["ewg_G_full6"], rtol=0, atol=1e-10)


def _h6_training():
    from evcont_amd.FCI_EVCont import FCI_EVCont_obj
    from evcont_amd.fci_small import SmallFCI
    from evcont_amd.hchain import hydrogen_chain
    cont = FCI_EVCont_obj(cisolver=SmallFCI(), cibasis="OAO")
    for d in (1.5, 2.0, 2.6):
        cont.append_to_rdms(hydrogen_chain(6, d, need_grad=False))
    return cont


@pytest.mark.parametrize("compress", [None, "sym8"])
def test_pyscf_md_branch_matches_native_integrator(compress):
    """get_trajectory on a Mole-shaped molecule runs ``pyscf.md.NVE`` (the stand-in reproduces its control flow) with
    the scanner as a GradScanner; the frames equal those of the native velocity-Verlet driver on the same chain.  The
    compressed layout requests int2e / int2e_ip1 from the Mole as aosym s4 / s2kl, written into the pinned staging
    buffers through ``out=``."""
    from evcont_amd.MD_utils import get_scanner, get_trajectory
    from evcont_amd.hchain import hydrogen_chain, s_gaussian_mol
    cont = _h6_training()
    R0 = hydrogen_chain(6, 1.9).atom_coords()
    v0 = 1e-4 * np.random.default_rng(3).standard_normal((6, 3))
    native = get_trajectory(hydrogen_chain(6, 1.9), cont.overlap, cont.one_rdm, cont.two_rdm, dt=8.0, steps=6,
                            init_veloc=v0, compress=compress)
    with pyscf_stub.installed() as ps:
        mol = pyscf_stub.StubMole(coords=R0, factory=lambda c: s_gaussian_mol(c))
        sc = get_scanner(mol, cont.one_rdm, cont.two_rdm, cont.overlap, compress=compress)
        assert isinstance(sc, ps.lib.GradScanner) and sc.converged is True
        traj = get_trajectory(mol.copy(), cont.overlap, cont.one_rdm, cont.two_rdm, dt=8.0, steps=6, init_veloc=v0,
                              compress=compress)
        # the queries of one step
        m2 = pyscf_stub.StubMole(coords=R0, factory=lambda c: s_gaussian_mol(c))
        sc2 = get_scanner(m2, cont.one_rdm, cont.two_rdm, cont.overlap, compress=compress)
        sc2(m2)
        big = {(q[0], q[1]) for q in m2.queries if q[0] in ("int2e", "int2e_ip1")}
    assert traj.shape == (6, 6, 3)
    np.testing.assert_allclose(traj, native, rtol=0, atol=1e-9)
    assert big == ({("int2e", "s4"), ("int2e_ip1", "s2kl")} if compress else {("int2e", "s1"), ("int2e_ip1", "s1")})


def test_integrator_callback_reads_predicted_rdm():
    """04_Zundel_continuation_MD.py:140-177: NVTBerendson(scanner, T, taut=..., callback=callback) with the callback
    reading ``locals["scanner"].base.predicted_one_rdm`` and ``locals["mol"]`` every step."""
    from evcont_amd.MD_utils import get_scanner
    from evcont_amd.ab_initio_gradients_loewdin import get_energy_with_grad
    from evcont_amd.hchain import hydrogen_chain, s_gaussian_mol
    cont = _h6_training()
    seen = []

    def callback(loc):
        D = loc["scanner"].base.predicted_one_rdm
        seen.append((loc["mol"].atom_coords(), np.array(D), float(np.trace(D))))

    with pyscf_stub.installed() as ps:
        mol = pyscf_stub.StubMole(coords=hydrogen_chain(6, 1.9).atom_coords(), factory=lambda c: s_gaussian_mol(c))
        mol.incore_anyway = True
        sc = get_scanner(mol, cont.one_rdm, cont.two_rdm, cont.overlap)
        assert sc.base.predicted_one_rdm is None and sc.base.predicted_two_rdm is None
        frames = []
        v0 = 2e-4 * np.random.default_rng(5).standard_normal((6, 3))
        ps.md.integrators.NVTBerendson(sc, 298.15, taut=250, steps=5, dt=6.0, incore_anyway=True, frames=frames,
                                       veloc=v0, callback=callback).run()
        G = sc.base.predicted_two_rdm      # the N^4 predicted 2-RDM is produced on demand, for the last geometry
    assert len(frames) == 5 and len(seen) == 5
    for R, D, tr in seen:
        assert abs(tr - 6.0) < 1e-9                                     # six electrons
        _, _, Dref, _ = get_energy_with_grad(s_gaussian_mol(R), cont.one_rdm, cont.two_rdm, cont.overlap,
                                             return_density_matrices=True)
        np.testing.assert_allclose(D, Dref, rtol=0, atol=1e-9)
    _, _, _, Gref = get_energy_with_grad(s_gaussian_mol(seen[-1][0]), cont.one_rdm, cont.two_rdm, cont.overlap,
                                         return_density_matrices=True)
    np.testing.assert_allclose(G, Gref, rtol=0, atol=1e-9)


def test_hosted_evaluator_matches_resident_path():
    """Pinned staging + graph replay (evcont_amd/hosted.py) against the plain evaluator, along a slowly varying
    sequence (eager priming calls, graph capture, replays) on both layouts."""
    import torch
    from evcont_amd.evaluator import ContinuationEvaluator, DeviceAO, DeviceTRDMs
    from evcont_amd.hosted import HostedEvaluator
    from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows
    from test_gpu_warm_start import blend
    dev = torch.device("cuda:0")
    n, T, A = 13, 5, 3
    S, one, two = make_trdms(n, T, 21)
    two_p = pack_rows(two, True, True)
    a0 = make_ao_arrays(n, A, 31, ao_sizes=(9, 2, 2), ip1_rs_symmetric=True)
    a1 = make_ao_arrays(n, A, 32, ao_sizes=(9, 2, 2), ip1_rs_symmetric=True)
    for comp, graph in ((None, True), ("sym8", False), ("sym8", True), (None, False)):
        trd = DeviceTRDMs(one, two_p, S, dev, compress=comp)
        hev = HostedEvaluator(trd, A, a0.aoslices, use_graph=graph)
        ref = ContinuationEvaluator(trd, A)
        assert hev.packed == (comp == "sym8")
        for k in range(7):
            ao = blend(a0, a1, 0.003 * k)
            E, g = hev.energy_with_grad(ao)
            Er, gr = ref.energy_with_grad(DeviceAO.from_arrays(ao, dev, pack_ip1=hev.packed, pack_eri=hev.packed))
            assert abs(E - Er) < 1e-11, (comp, k)
            np.testing.assert_allclose(g, gr, rtol=0, atol=1e-10)
        assert (hev.graph is not None) == graph


@pytest.mark.parametrize("script,args", [
    ("h10_forces.py", ["--radius", "0.2", "--points", "12", "--exact", "0", "--fixture"]),
    ("h30_md.py", ["--atoms", "8", "--train", "4", "--steps", "4"]),
    ("zundel_md.py", ["--demo", "--steps", "4"]),
    ("h2o_md.py", ["--demo", "--steps", "3"]),
])
def test_example_drivers_run(script, args, tmp_path):
    r = subprocess.run([sys.executable, os.path.join(REPO, "examples", script)] + args, cwd=tmp_path,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "OK" in r.stdout
