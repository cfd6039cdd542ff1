#!/usr/bin/env python3
"""NVE molecular dynamics of a hydrogen chain on the continuation surface -- the evaluator-facing part of the
reference's ``scripts/MD/H30/md_H30_evcont_from_DMRG.py`` (BASELINE configs[2]): a container whose
``overlap / one_rdm / two_rdm`` are loaded from ``overlap.npy, one_rdm.npy, two_rdm.npy`` when those files exist
(the reference's resume protocol, lines 72-85), ``get_trajectory`` (= ``get_scanner`` + integrator) on H_n / STO-6G
started at 1.9 Bohr spacing with dt = 5, and the trajectory written as ``traj_EVCont_0.npy``.

What is NOT here is the training itself: the reference trains with DMRG (block2, not in this image).  Without the
``.npy`` files the driver therefore builds a SEEDED SYNTHETIC training set of the right shapes and symmetries
(``--train`` states) -- the numbers along the trajectory are then not physics, the data flow and its cost are what the
script shows: per step one host integral evaluation (closed-form s-Gaussians here, PySCF/libcint in the reference),
one staged upload (12 MB at H30 with the compressed layout) and one graph launch on the device.

    python examples/h30_md.py [--atoms 30] [--train 20] [--steps 5] [--dt 5]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from evcont_amd.DMRG_EVCont import DMRG_EVCont_obj                                 # noqa: E402
from evcont_amd.MD_utils import get_scanner, nve_velocity_verlet                   # noqa: E402
from evcont_amd.hchain import s_gaussian_mol, STO3G_H_EXPONENTS, STO3G_H_COEFFICIENTS, \
    STO6G_H_EXPONENTS, STO6G_H_COEFFICIENTS                                        # noqa: E402
from evcont_amd.synthetic import make_trdms, pack_rows                             # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--atoms", type=int, default=30)
p.add_argument("--train", type=int, default=20)
p.add_argument("--steps", type=int, default=5)
p.add_argument("--dt", type=float, default=5.0)
a = p.parse_args()
norb = nelec = a.atoms
init_dist = 1.9


# (the host-side integral generator of this repository keeps the primitive ERIs in memory: STO-6G like the reference
#  up to 12 atoms, STO-3G -- the same number of orbitals -- beyond; ~13 s per H30 geometry on 8 cores, which is why
#  the default number of steps is small: the continuation itself is what the device does)
_ex, _co = (STO6G_H_EXPONENTS, STO6G_H_COEFFICIENTS) if a.atoms <= 12 else (STO3G_H_EXPONENTS, STO3G_H_COEFFICIENTS)


def get_mol(geometry):
    return s_gaussian_mol(np.asarray(geometry), exponents=_ex, coefficients=_co)


init_mol = get_mol(np.array([[0, 0, init_dist * i] for i in range(nelec)]))
cont = DMRG_EVCont_obj(dmrg_converge_fun=None, append_method=None)
if os.path.exists("overlap.npy"):
    cont.overlap, cont.one_rdm, cont.two_rdm = np.load("overlap.npy"), np.load("one_rdm.npy"), np.load("two_rdm.npy")
    print(f"training data loaded: {cont.overlap.shape[0]} states, two_rdm {cont.two_rdm.shape}")
else:
    # stand-in with the symmetries every container of the reference produces, in the pair-packed layout
    if norb <= 12:
        S, one, two = make_trdms(norb, a.train, 7)
        two = pack_rows(two, True, True)
    else:   # (T,T,N,N,N,N) would be 2.6 GB at H30: draw the packed rows directly
        rng = np.random.default_rng(7)
        T = a.train
        A = rng.standard_normal((T, T))
        S = A @ A.T / T + np.eye(T)
        d = rng.standard_normal((T, T, norb, norb)) / norb
        one = 0.5 * (d + d.transpose(1, 0, 3, 2))
        n2 = norb * norb
        two = rng.standard_normal((T * (T + 1) // 2, n2 * (n2 + 1) // 2)) / n2
    cont.overlap, cont.one_rdm, cont.two_rdm = S, one, two
    print(f"no overlap.npy here: synthetic stand-in for the DMRG training set ({a.train} states)")

scanner = get_scanner(init_mol, cont.one_rdm, cont.two_rdm, cont.overlap, compress="sym8")
t_host = [0.0]
_with = init_mol.with_coords


def timed_with_coords(R):                      # where the reference calls libcint
    t0 = time.time()
    m = _with(R)
    t_host[0] += time.time() - t0
    return m


init_mol.with_coords = timed_with_coords
t0 = time.time()
frames = nve_velocity_verlet(scanner, init_mol, dt=a.dt, steps=a.steps)
wall = time.time() - t0
traj = np.array([f["coord"] for f in frames])
np.save("traj_EVCont_0.npy", traj)
etot = np.array([f["epot"] + f["ekin"] for f in frames])
print(f"{a.steps} steps of H{a.atoms}: {wall:.2f} s wall, of which host integrals {t_host[0]:.2f} s; "
      f"continuation steps/s excluding them: {a.steps / max(wall - t_host[0], 1e-9):.0f}; "
      f"total-energy drift {etot[-1] - etot[0]:.2e} Ha")
assert np.all(np.isfinite(traj)) and np.all(np.isfinite(etot))
print("OK")
