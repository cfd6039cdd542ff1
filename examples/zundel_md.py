#!/usr/bin/env python3
"""Thermostatted MD with observables from the predicted 1-RDM -- the flow of the reference's
``scripts/MD/Zundel_thermodynamics/continuation/04_Zundel_continuation_MD.py`` (BASELINE configs[4]):

    per-pair files  ../MPS_cross_{i}_{j}/{ovlp,one_rdm,two_rdm}.npy   (lines 99-128)
      -> training data resident on the device          (here: streamed straight into the packed device layout by
                                                         ``trdm_io.load_pair_directories``, never assembled on the host)
      -> ``get_scanner(init_mol, ...)``                 (line 131)
      -> NVT-Berendsen integrator with ``callback=``    (lines 140-177), the callback reading
         ``locals["scanner"].base.predicted_one_rdm`` and back-rotating it to the AO basis for dipole moment / charges.

With PySCF installed, pass ``--xyz`` / ``--pairs`` for a real molecule (integrals by ``mol.intor`` written into the
pinned staging buffers, integrator ``pyscf.md.integrators.NVTBerendson``).  ``--demo`` runs the same flow end to end in
this image: an H6 chain (closed-form integrals, FCI training states from ``evcont_amd.fci_small``), the per-pair
directories written by ``trdm_io.save_pair_directories`` first, the native integrator with the same callback contract.

    python examples/zundel_md.py --demo [--steps 20]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from evcont_amd import trdm_io                                                     # noqa: E402
from evcont_amd.MD_utils import get_scanner, nve_velocity_verlet                   # noqa: E402
from evcont_amd.ab_initio_eigenvector_continuation import get_basis                # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--demo", action="store_true")
p.add_argument("--pairs", default="..", help="directory holding the MPS_cross_i_j directories")
p.add_argument("--ntrain", type=int, default=0)
p.add_argument("--xyz", default=None)
p.add_argument("--steps", type=int, default=20)
p.add_argument("--dt", type=float, default=5.0)
p.add_argument("--temperature", type=float, default=298.15)
a = p.parse_args()
rng = np.random.default_rng(1)

if a.demo:
    from evcont_amd.FCI_EVCont import FCI_EVCont_obj
    from evcont_amd.fci_small import SmallFCI
    from evcont_amd.hchain import hydrogen_chain
    from evcont_amd.synthetic import pack_rows
    cont = FCI_EVCont_obj(cisolver=SmallFCI(), cibasis="OAO")
    for d in (1.5, 1.9, 2.4, 3.0):
        cont.append_to_rdms(hydrogen_chain(6, d, need_grad=False))
    a.pairs, a.ntrain = "pairs_demo", cont.overlap.shape[0]
    trdm_io.save_pair_directories(a.pairs, cont.overlap, cont.one_rdm, pack_rows(cont.two_rdm, True, True))
    init_mol = hydrogen_chain(6, 2.0)
    charges = np.ones(6)
else:
    from pyscf import gto
    init_mol = gto.M(atom=a.xyz, basis="6-31G", unit="Angstrom", charge=1)
    charges = init_mol.atom_charges()

# training data: per-pair files -> device (packed (P, M) rows, compressed there)
trd = trdm_io.load_pair_directories(a.pairs, a.ntrain).compress_sym8_()
scanner_fun = get_scanner(init_mol, None, None, None, hermitian=True, device_trdms=trd)

open("dipole_moment_continuation.txt", "w").close()
open("atom_charges_continuation.txt", "w").close()


def callback(locals):
    mol = locals["mol"]
    basis = get_basis(mol)
    predicted_one_rdm = locals["scanner"].base.predicted_one_rdm
    predicted_one_rdm_ao = basis.dot(predicted_one_rdm).dot(basis.T)
    S = mol.S if hasattr(mol, "S") else mol.intor("int1e_ovlp")
    # Mulliken populations per atom (hf.mulliken_meta in the reference) and the nuclear part of the dipole
    pop = np.einsum("ij,ji->i", predicted_one_rdm_ao, S)
    sl = np.asarray(mol.aoslices) if hasattr(mol, "aoslices") else np.array([s[2:] for s in mol.aoslice_by_atom()])
    atomic_charges = charges - np.array([pop[s0:s1].sum() for s0, s1 in sl])
    coords = mol.atom_coords()
    nucl_dip = np.einsum("i,ix->x", charges, coords - coords.mean(axis=0))
    with open("dipole_moment_continuation.txt", "a") as fl:
        fl.write("  ".join(str(x) for x in nucl_dip) + "\n")
    with open("atom_charges_continuation.txt", "a") as fl:
        fl.write("  ".join(str(x) for x in atomic_charges) + "\n")
    assert abs(pop.sum() - predicted_one_rdm.trace()) < 1e-8


natm = len(charges)
mass = np.asarray(init_mol.atom_mass_list()) * 1822.888486209
veloc = rng.standard_normal((natm, 3)) * np.sqrt(3.166811563e-6 * a.temperature / mass)[:, None]   # Maxwell-Boltzmann
if a.demo:
    frames = nve_velocity_verlet(scanner_fun, init_mol, dt=a.dt, steps=a.steps, veloc=veloc, callback=callback,
                                 thermostat=(a.temperature, 250.0), trajectory_output="trajectory.xyz")
    traj = np.array([f["coord"] for f in frames])
else:
    from pyscf import md
    frames = []
    scanner_fun.mol = init_mol.copy()
    md.integrators.NVTBerendson(scanner_fun, a.temperature, taut=250, steps=a.steps, dt=a.dt, incore_anyway=True,
                                frames=frames, veloc=veloc, trajectory_output="trajectory.xyz",
                                data_output="energy.xyz", callback=callback).run()
    traj = np.array([frame.coord for frame in frames])
np.save("trajectory.npy", traj)
q = np.loadtxt("atom_charges_continuation.txt")
assert q.shape == (a.steps, natm) and np.all(np.isfinite(traj))
print(f"{a.steps} steps, {natm} atoms; net charge along the run {q.sum(axis=1).min():+.2e} .. {q.sum(axis=1).max():+.2e}")
print("OK")
