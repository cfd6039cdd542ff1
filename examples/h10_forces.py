#!/usr/bin/env python3
"""Energies and forces of 1000 randomly distorted H10 chains from a continuation trained on five symmetrically
stretched chains -- the continuation part of the reference's
``scripts/PES_H_chain/H10_PES/H10_continuation_3D_replacements.py`` (Fig. 2; BASELINE configs[1]) on the MI355X path.

Same protocol: equilibrium spacing 1.78596 Bohr, training stretches 0, +-0.5, +-1.0 Bohr, every atom displaced by
`radius` in a random direction, ``default_rng(seed=1)``, same files (``H10_predicted_energies_<radius>.txt``,
``H10_continuation_gradients_<radius>.txt``).  Without PySCF in this image the molecule is the closed-form STO-6G
s-Gaussian chain of ``evcont_amd.hchain`` and the FCI solver ``evcont_amd.fci_small`` (the HF / CASCI / GAP columns
of the reference need PySCF / dscribe and are left out; ``--exact N`` adds this repository's FCI energy for the first
N geometries).  The 1000 evaluations run (a) through the reference-shaped call ``get_energy_with_grad`` for the first
few and (b) as batched device calls over all of them, which is how a scan should be run on this hardware.

    python examples/h10_forces.py [--radius 0.2] [--points 1000] [--exact 3] [--fixture]
"""
import argparse
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from evcont_amd.FCI_EVCont import FCI_EVCont_obj                                   # noqa: E402
from evcont_amd.ab_initio_gradients_loewdin import get_energy_with_grad            # noqa: E402
from evcont_amd.electron_integral_utils import get_basis, get_integrals           # noqa: E402
from evcont_amd.evaluator import BatchedEvaluator, DeviceAOBatch                   # noqa: E402
from evcont_amd.fci_small import SmallFCI                                          # noqa: E402
from evcont_amd.hchain import s_gaussian_mol, STO3G_H_EXPONENTS, STO3G_H_COEFFICIENTS, \
    STO6G_H_EXPONENTS, STO6G_H_COEFFICIENTS                                        # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--radius", type=float, default=0.2)
p.add_argument("--points", type=int, default=1000)
p.add_argument("--exact", type=int, default=3, help="FCI reference energies for the first N geometries")
p.add_argument("--batch", type=int, default=50)
p.add_argument("--fixture", action="store_true",
               help="take the five training states from tests/golden/h10_fci_t5.npz (STO-3G, other spacings) instead "
                    "of solving five 63504-determinant FCI problems first (~2 min)")
a = p.parse_args()

n_data_points, seed, natm = a.points, 1, 10
rng = np.random.default_rng(seed)
equilibrium_dist = 1.78596
equilibrium_pos = np.array([(x * equilibrium_dist, 0.0, 0.0) for x in range(natm)])
ex, co = (STO3G_H_EXPONENTS, STO3G_H_COEFFICIENTS) if a.fixture else (STO6G_H_EXPONENTS, STO6G_H_COEFFICIENTS)


def get_mol(positions, need_grad=True):
    return s_gaussian_mol(np.asarray(positions), exponents=ex, coefficients=co, need_grad=need_grad)


continuation_object = FCI_EVCont_obj(cisolver=SmallFCI(), cibasis="OAO")
if a.fixture:
    with np.load(os.path.join(REPO, "tests", "golden", "h10_fci_t5.npz")) as z:
        continuation_object.overlap, continuation_object.one_rdm = z["overlap"], z["one_rdm"]
        continuation_object.two_rdm = z["two_rdm_pack2"]          # (P, M): the reference's packed layout
else:
    for dist in equilibrium_dist + np.array([0.0, 0.5, -0.5, 1.0, -1.0]):
        t0 = time.time()
        continuation_object.append_to_rdms(get_mol([(x, 0.0, 0.0) for x in dist * np.arange(natm)], need_grad=False))
        print(f"training point d = {dist:.5f}: E_FCI = {continuation_object.ens[-1]:.10f}  ({time.time() - t0:.0f} s)",
              flush=True)

# the reference's sampling loop (same generator calls in the same order)
geoms = []
for i in range(n_data_points):
    theta = rng.random(size=(natm)) * np.pi
    phi = rng.random(size=(natm)) * 2 * np.pi
    disp = np.stack((a.radius * np.sin(theta) * np.cos(phi), a.radius * np.sin(theta) * np.sin(phi),
                     a.radius * np.cos(theta)), axis=-1)
    geoms.append(equilibrium_pos + disp)

one, two, S = continuation_object.one_rdm, continuation_object.two_rdm, continuation_object.overlap
# (a) the reference's call, geometry by geometry
first = [get_energy_with_grad(get_mol(R), one, two, S) for R in geoms[:3]]

# (b) all geometries in batched device calls against the resident compressed t-RDMs
from evcont_amd.evaluator import DeviceTRDMs, DeviceAO                             # noqa: E402
trd = DeviceTRDMs(one, two, S, compress="sym8")
G = min(a.batch, n_data_points)
be = BatchedEvaluator(trd, natm, G)
en = np.zeros(n_data_points)
gr = np.zeros((n_data_points, natm, 3))
t_int = t_dev = 0.0
for b0 in range(0, n_data_points, G):
    idx = [min(b0 + k, n_data_points - 1) for k in range(G)]          # (the last batch is padded)
    t0 = time.time()
    mols = [get_mol(geoms[i]) for i in idx]
    t1 = time.time()
    E, g = be.energies_with_grads(DeviceAOBatch.stack([DeviceAO.from_arrays(m, trd.device, pack_ip1=True, pack_eri=True)
                                                       for m in mols]))
    t_int, t_dev = t_int + t1 - t0, t_dev + time.time() - t1
    for k, i in enumerate(idx):
        en[i], gr[i] = E[k], g[k]
for k, (E, g) in enumerate(first):
    assert abs(E - en[k]) < 1e-9 and np.abs(g - gr[k]).max() < 1e-8, (k, E, en[k])

exact = np.full(n_data_points, np.nan)
solver = SmallFCI()
for i in range(min(a.exact, n_data_points)):
    m = get_mol(geoms[i], need_grad=False)
    h1, h2 = get_integrals(m, get_basis(m))
    exact[i] = solver.kernel(h1, h2, m.nao, m.nelec)[0] + m.energy_nuc()
    assert en[i] > exact[i] - 1e-9          # variational

with open("H10_predicted_energies_{}.txt".format(a.radius), "w") as fl:
    fl.write("FCI  Continuation\n")
    for i in range(n_data_points):
        fl.write("{}  {}\n".format(exact[i], en[i]))
with open("H10_continuation_gradients_{}.txt".format(a.radius), "w") as fl:
    for i in range(n_data_points):
        for row in gr[i]:
            fl.write("{}  {}  {}  ".format(*row))
        fl.write("\n")
print(f"{n_data_points} geometries: host integrals {t_int:.2f} s, upload + device + download {t_dev:.2f} s; "
      f"E[0] = {en[0]:.10f}, max |force| = {np.abs(gr).max():.6f}")
print("OK")
