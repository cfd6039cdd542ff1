#!/usr/bin/env python3
"""PES of a 6-atom hydrogen chain from 1, 2 and 3 FCI training points — the flow of the reference's
``scripts/PES_H_chain/H6_PES/H6_continuation.py`` (BASELINE configs[0]) on the MI355X path, without
PySCF: closed-form s-Gaussian integrals (``evcont_amd.hchain``, STO-6G like the reference script),
FCI training states from ``evcont_amd.fci_small`` held in ``FCI_EVCont_obj``, and the scan evaluated
(a) geometry by geometry through the reference-shaped call ``approximate_ground_state_OAO`` and
(b) as ONE batched device call over all 50 geometries (``BatchedEvaluator``), which is how a scan
should be run on this hardware.

    python examples/h6_pes.py            # needs a HIP device; writes predicted_surface_*.txt, exact_surface.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from evcont_amd.FCI_EVCont import FCI_EVCont_obj                                   # noqa: E402
from evcont_amd.ab_initio_eigenvector_continuation import approximate_ground_state_OAO  # noqa: E402
from evcont_amd.electron_integral_utils import get_basis, get_integrals           # noqa: E402
from evcont_amd.evaluator import BatchedEvaluator, DeviceAO, DeviceAOBatch        # noqa: E402
from evcont_amd.fci_small import SmallFCI                                          # noqa: E402
from evcont_amd.hchain import s_gaussian_mol, STO6G_H_EXPONENTS, STO6G_H_COEFFICIENTS  # noqa: E402

n_atoms = 6


def get_mol(dist, need_grad=False):
    x = (np.arange(n_atoms) - np.median(np.arange(n_atoms))) * dist
    coords = np.stack([x, np.zeros(n_atoms), np.zeros(n_atoms)], axis=1)
    return s_gaussian_mol(coords, exponents=STO6G_H_EXPONENTS, coefficients=STO6G_H_COEFFICIENTS,
                          need_grad=need_grad)


def main():
    solver = SmallFCI()
    continuation_object = FCI_EVCont_obj(cisolver=solver, cibasis="OAO")
    test_dists = np.linspace(0.8, 3.0)
    test_mols = [get_mol(d) for d in test_dists]
    for i, trn_dist in enumerate([1.0, 1.8, 2.6]):
        continuation_object.append_to_rdms(get_mol(trn_dist))
        # (a) the reference's call, one geometry at a time
        ens = [approximate_ground_state_OAO(mol, continuation_object.one_rdm, continuation_object.two_rdm,
                                            continuation_object.overlap)[0] for mol in test_mols]
        # (b) the whole scan as one batched device call against the container's resident packed t-RDMs
        trd = continuation_object.device_trdms()
        be = BatchedEvaluator(trd, n_atoms, len(test_mols))
        be.enqueue(DeviceAOBatch.stack([DeviceAO.from_arrays(m, trd.device, energy_only=True) for m in test_mols]),
                   energy_only=True)
        be.synchronize()
        ens_batched = be.energy[:, 0].cpu().numpy()
        assert np.abs(ens_batched - np.array(ens)).max() < 1e-10
        np.savetxt(f"predicted_surface_{i + 1}_datapoints.txt", np.stack([test_dists, ens], axis=1))
        print(f"{i + 1} training point(s): E(2.2) = {np.interp(2.2, test_dists, ens):.8f}")
    exact = []
    for mol in test_mols:
        h1, h2 = get_integrals(mol, get_basis(mol))
        exact.append(solver.kernel(h1, h2, mol.nao, mol.nelec)[0] + mol.energy_nuc())
    np.savetxt("exact_surface.txt", np.stack([test_dists, exact], axis=1))
    err = np.array(ens) - np.array(exact)
    print(f"3 training points: max error {err.max():.2e} Ha, min error {err.min():.2e} Ha (variational: >= 0)")


if __name__ == "__main__":
    main()
