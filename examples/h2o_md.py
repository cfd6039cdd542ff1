#!/usr/bin/env python3
"""MD of a water molecule on the continuation surface of CASCI training states -- the evaluator-facing part of the
reference's ``scripts/MD/H2O/md_H2O_*_CAS_continuation.py`` (BASELINE configs[3]: 6-31G, N = 13; the cc-pVTZ variant
``md_H2O_vtz_CAS_continuation.py:25-33``: N = 58, the reference's largest orbital space):

    CAS_EVCont_obj(ncas, neleca)  ->  converge_EVCont_MD(container, init_mol, steps, dt, ...)

With PySCF (and pygnme for the CASCI transition RDMs) installed, ``--train`` runs exactly that call: RHF + CASCI training
states on the host, the MD inner loop -- one ``scanner(mol)`` per step -- on the device.  Neither is in this image, so
``--demo`` shows the part this repository replaces, end to end and at the right sizes: a container of ``--states``
training states with the shapes and symmetries the CAS container produces (seeded stand-in, or ``overlap.npy /
one_rdm.npy / two_rdm.npy`` if they are there: the reference's resume files), a water-SHAPED array-level molecule
(three atoms, AO slices 9/2/2 for 6-31G or 30/14/14 for cc-pVTZ) whose integrals are seeded tensors with the index
symmetries of real ones that vary smoothly with the geometry, and ``get_scanner`` + velocity Verlet on it.  The numbers
along such a trajectory are not physics; the data flow and its cost per step are what the script shows: at cc-pVTZ size
one step moves 162 MB of packed integrals to the device and runs the 64-wide symmetric pipeline (csrc/pair64.hip).

    python examples/h2o_md.py --demo [--basis 6-31G|cc-pVTZ] [--states 6] [--steps 5] [--dt 5]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from evcont_amd.CASCI_EVCont import CAS_EVCont_obj                                  # noqa: E402
from evcont_amd.MD_utils import converge_EVCont_MD, get_scanner, nve_velocity_verlet   # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--demo", action="store_true")
p.add_argument("--train", action="store_true", help="the reference's active-learning run (needs PySCF and pygnme)")
p.add_argument("--basis", default="6-31G", choices=["6-31G", "cc-pVTZ"])
p.add_argument("--states", type=int, default=6)
p.add_argument("--steps", type=int, default=5)
p.add_argument("--dt", type=float, default=5.0)
a = p.parse_args()

ncas, neleca = 8, 4                     # active space of the reference scripts
a_to_bohr, stretch_factor = 1.8897259886, 1.2
init_geometry = a_to_bohr * stretch_factor * np.array([[0.0, 0.795, -0.454], [0.0, -0.795, -0.454], [0.0, 0.0, 0.113]])

if a.train:
    from pyscf import gto               # (ImportError here: this is the branch that needs the host-side chemistry)

    def get_mol(geometry):
        mol = gto.Mole()
        mol.build(atom=[("H", geometry[0]), ("H", geometry[1]), ("O", geometry[2])], basis=a.basis, symmetry=False,
                  unit="Bohr")
        return mol

    converge_EVCont_MD(CAS_EVCont_obj(ncas, neleca), get_mol(init_geometry), steps=a.steps, dt=a.dt,
                       prune_irrelevant_data=False, data_addition="farthest_point_ham")
    print("OK")
    sys.exit(0)

if not a.demo:
    p.error("pass --demo (this image) or --train (PySCF + pygnme)")

from evcont_amd.synthetic import AOArrays, make_ao_arrays                           # noqa: E402

sizes = (2, 2, 9) if a.basis == "6-31G" else (14, 14, 30)      # H, H, O
n, T = sum(sizes), a.states
masses = np.array([1.008, 1.008, 15.999])


class WaterShaped:
    """Three atoms whose AO integrals are a smooth function of the coordinates: a blend of two seeded integral sets
    (``evcont_amd.synthetic.make_ao_arrays``: symmetric positive definite overlap, 8-fold symmetric ``int2e``,
    ``int2e_ip1`` symmetric in its last two indices) weighted by the mean displacement from the start geometry."""

    def __init__(self, coords, ends=None):
        self.coords = np.array(coords, dtype=np.float64)
        self.ends = ends or tuple(make_ao_arrays(n, 3, 4100 + k, ao_sizes=sizes, ip1_rs_symmetric=True) for k in (0, 1))
        w = float(np.tanh(np.abs(self.coords - init_geometry).mean()))
        mix = lambda x, y: (1.0 - w) * np.asarray(x) + w * np.asarray(y)
        e0, e1 = self.ends
        self.ao = AOArrays(mix(e0.S, e1.S), mix(e0.hcore, e1.hcore), mix(e0.eri, e1.eri), mix(e0.ipovlp, e1.ipovlp),
                           mix(e0.dhcore, e1.dhcore), mix(e0.eri_ip1, e1.eri_ip1), e0.aoslices,
                           float(mix(e0.enuc, e1.enuc)), mix(e0.gnuc, e1.gnuc), integral_symmetry=True)

    def __getattr__(self, name):            # S, hcore, eri, ... : the array-level ``mol`` queries
        return getattr(self.__dict__["ao"], name)

    def atom_coords(self):
        return self.coords

    def atom_mass_list(self):
        return masses

    def with_coords(self, R):
        return WaterShaped(R, self.ends)


cont = CAS_EVCont_obj(ncas, neleca)
if os.path.exists("overlap.npy"):
    cont.overlap, cont.one_rdm, cont.two_rdm = np.load("overlap.npy"), np.load("one_rdm.npy"), np.load("two_rdm.npy")
    print(f"training data loaded: {cont.overlap.shape[0]} states, two_rdm {cont.two_rdm.shape}")
else:
    # stand-in with the symmetries the container's transition RDMs have, drawn directly in the pair-packed layout
    # ((T, T, N, N, N, N) would be 3.3 GB at cc-pVTZ size with six states)
    rng = np.random.default_rng(11)
    A = rng.standard_normal((T, T))
    S = A @ A.T / T + np.eye(T)
    d = rng.standard_normal((T, T, n, n)) / n
    one = 0.5 * (d + d.transpose(1, 0, 3, 2))
    n2 = n * n
    two = rng.standard_normal((T * (T + 1) // 2, n2 * (n2 + 1) // 2)) / n2
    cont.overlap, cont.one_rdm, cont.two_rdm = S, one, two
    print(f"no overlap.npy here: synthetic stand-in for {T} CASCI training states, N = {n} ({a.basis} water shape)")

init_mol = WaterShaped(init_geometry)
t0 = time.time()
scanner = get_scanner(init_mol, cont.one_rdm, cont.two_rdm, cont.overlap)      # default arguments, as the reference calls it
e0, g0 = scanner(init_mol)
t_first = time.time() - t0
t_host = [0.0]
_with = init_mol.with_coords


def timed_with_coords(R):                  # where the reference calls libcint
    t1 = time.time()
    m = _with(R)
    t_host[0] += time.time() - t1
    return m


init_mol.with_coords = timed_with_coords
t0 = time.time()
frames = nve_velocity_verlet(scanner, init_mol, dt=a.dt, steps=a.steps)
wall = time.time() - t0
traj = np.array([f["coord"] for f in frames])
np.save("traj_EVCont_0.npy", traj)
print(f"first call (upload, compression, set-up) {t_first:.2f} s; {a.steps} steps: {wall:.2f} s wall, of which the "
      f"stand-in integrals {t_host[0]:.2f} s; continuation steps/s excluding them: "
      f"{a.steps / max(wall - t_host[0], 1e-9):.0f}; compressed layout on the device: "
      f"{bool(getattr(scanner, '_hev', None) is not None and scanner._hev.packed)}")
assert np.all(np.isfinite(traj)) and np.isfinite(e0) and np.all(np.isfinite(g0))
print("OK")
