#!/usr/bin/env python3
"""Generate the golden vectors in this directory by running the REFERENCE's own
functions (BoothGroup/evcont, mounted read-only at /root/reference) on seeded
array-level inputs.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py

How the reference is driven: its hot-path modules
do ``from pyscf import ...`` at import time, and PySCF is not installed here.
An *empty* ``pyscf`` namespace is registered so the imports succeed; the handful
of leaf calls the ``mol``-taking functions make are answered from seeded arrays
held by ``FakeMol`` (``mol.intor``, ``scf.hf.get_hcore``,
``grad.RHF(...).grad_nuc/hcore_generator``) and ``ao2mo.kernel`` is answered by
its published definition, the dense four-index transformation.  All arithmetic
that is recorded as "expected output" is executed by the reference's code.
Nothing from the reference is copied: this script only imports and calls it.

Outputs: ``tests/golden/case_*.npz`` (inputs + expected outputs, float64).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

REFERENCE = os.environ.get("EVCONT_REFERENCE", "/root/reference")


def install_namespace():
    ps = types.ModuleType("pyscf")
    subs = {n: types.ModuleType("pyscf." + n) for n in ("scf", "lo", "ao2mo", "grad", "fci", "md", "lib")}
    for n, m in subs.items():
        setattr(ps, n, m)
        sys.modules["pyscf." + n] = m
    sys.modules["pyscf"] = ps
    subs["scf"].hf = types.SimpleNamespace(get_hcore=lambda mol: mol._hcore)
    subs["scf"].RHF = lambda mol: mol

    class _Grad:
        def __init__(self, mol):
            self.mol = mol

        def grad_nuc(self):
            return self.mol._gnuc

        def hcore_generator(self):
            return lambda i: self.mol._dhcore[i]

    subs["grad"].RHF = _Grad
    subs["ao2mo"].kernel = lambda mol, C: np.einsum(
        "ijkl,ia,jb,kc,ld->abcd", mol._eri, C, C, C, C, optimize=True)
    subs["ao2mo"].restore = lambda sym, eri, n: np.asarray(eri).reshape(n, n, n, n)
    subs["lib"].GradScanner = type("GradScanner", (), {})
    mpi = types.ModuleType("mpi4py")
    mpi.MPI = types.SimpleNamespace(COMM_WORLD=types.SimpleNamespace(
        Get_rank=lambda: 0, Get_size=lambda: 1, rank=0))
    sys.modules["mpi4py"] = mpi


class FakeMol:
    def __init__(self, ao):
        self.nao = ao.nao
        self.natm = ao.natm
        self._S, self._hcore, self._eri = ao.S, ao.hcore, ao.eri
        self._ipovlp, self._dhcore, self._ip1 = ao.ipovlp, ao.dhcore, ao.eri_ip1
        self._enuc, self._gnuc = ao.enuc, ao.gnuc
        self._sl = [(0, 0, int(a), int(b)) for a, b in ao.aoslices]

    def intor(self, name, comp=None):
        return {"int1e_ovlp": self._S, "int1e_ipovlp": self._ipovlp,
                "int2e": self._eri, "int2e_ip1": self._ip1}[name].copy()

    def aoslice_by_atom(self):
        return self._sl

    def energy_nuc(self):
        return self._enuc


CASES = [
    # name, N, T, A, ao_sizes, degenerate_S, seed
    ("n4t2a2", 4, 2, 2, None, False, 11),
    ("n6t3a3", 6, 3, 3, None, False, 12),
    ("n6t3a6", 6, 3, 6, None, False, 13),
    ("n8t2a3_degS", 8, 2, 3, (3, 3, 2), True, 14),
    ("n5t4a2", 5, 4, 2, (2, 3), False, 15),
]


def main():
    install_namespace()
    sys.path.insert(0, REFERENCE)
    import evcont.electron_integral_utils as eiu
    import evcont.ab_initio_eigenvector_continuation as evc
    import evcont.ab_initio_gradients_loewdin as gl

    from evcont_amd.synthetic import make_ao_arrays, make_trdms, pack_rows

    captured = {}
    real_eigh, real_eig = evc.eigh, evc.eig

    def spy_eigh(H, S):
        captured["H"] = np.array(H, copy=True)
        return real_eigh(H, S)

    def spy_eig(H, S):
        captured["H"] = np.array(H, copy=True)
        return real_eig(H, S)

    evc.eigh, evc.eig = spy_eigh, spy_eig

    for name, n, T, A, sizes, degS, seed in CASES:
        ao = make_ao_arrays(n, A, seed, ao_sizes=sizes, degenerate_S=degS)
        S_train, one, two = make_trdms(n, T, seed + 100)
        mol = FakeMol(ao)
        out = dict(S=ao.S, hcore=ao.hcore, eri=ao.eri, ipovlp=ao.ipovlp, dhcore=ao.dhcore,
                   eri_ip1=ao.eri_ip1, aoslices=ao.aoslices, enuc=ao.enuc, gnuc=ao.gnuc,
                   S_train=S_train, one_RDM=one, two_RDM=two)

        X = eiu.get_loewdin_trafo(ao.S.copy())
        h1, h2 = eiu.get_integrals(mol, X)
        out.update(X=X, h1=h1, h2=h2)
        out["h2_packed_half"] = eiu.compress_electron_exchange_symmetry(h2.copy(), diag_multiplier=0.5)
        out["h2_packed_one"] = eiu.compress_electron_exchange_symmetry(h2.copy())
        out["h2_restored"] = eiu.restore_electron_exchange_symmetry(out["h2_packed_one"], n)

        layouts = {"full6": two,
                   "pair5": pack_rows(two, True, False),
                   "elec3": pack_rows(two, False, True),
                   "pack2": pack_rows(two, True, True)}
        for lname, g in layouts.items():
            for herm in (True, False):
                tag = f"{lname}_{'h' if herm else 'nh'}"
                e, c = evc.approximate_ground_state(h1, h2.copy(), one, g, S_train, hermitian=herm)
                out[f"gs_E_{tag}"] = e
                out[f"gs_c_{tag}"] = c
                out[f"gs_H_{tag}"] = captured["H"]
                nroots = min(T, 3)
                em, cm = evc.approximate_multistate(h1, h2.copy(), one, g, S_train,
                                                    nroots=nroots, hermitian=herm)
                out[f"ms_E_{tag}"] = em
                out[f"ms_C_{tag}"] = cm
                et, ct = evc.approximate_ground_state_OAO(mol, one, g, S_train, hermitian=herm)
                out[f"gsoao_E_{tag}"] = et
            E, grad, D, G = gl.get_energy_with_grad(mol, one, g, S_train, hermitian=True,
                                                    return_density_matrices=True)
            out[f"ewg_E_{lname}"] = E
            out[f"ewg_grad_{lname}"] = grad
            out[f"ewg_D_{lname}"] = D
            out[f"ewg_G_{lname}"] = G
        # non-Hermitian energy+grad (the reference's eig branch returns 2-norm-normalised vectors)
        E, grad, D, G = gl.get_energy_with_grad(mol, one, two, S_train, hermitian=False,
                                                return_density_matrices=True)
        out.update(ewg_E_full6_nh=E, ewg_grad_full6_nh=grad, ewg_D_full6_nh=D, ewg_G_full6_nh=G)

        # gradient building blocks
        out["dS"] = gl.get_overlap_grad(mol)
        out["LG"] = gl.loewdin_trafo_grad(ao.S.copy())
        dX = gl.get_derivative_ao_mo_trafo(mol)
        out["dX"] = dX
        out["h1_jac_ao"] = gl.get_one_el_grad_ao(mol)
        out["h1_jac"] = gl.get_one_el_grad(mol, ao_mo_trafo=X, ao_mo_trafo_grad=dX)
        out["h1_jac_default"] = gl.get_one_el_grad(mol)
        D = out["ewg_D_full6"]
        G = out["ewg_G_full6"]
        slices = tuple((int(a), int(b)) for a, b in ao.aoslices)
        out["two_el_grad"] = gl.two_el_grad(ao.eri.copy(), G, X, dX, ao.eri_ip1.copy(), slices)
        out["grad_elec"] = gl.get_grad_elec_OAO(mol, D, G, ao_mo_trafo=X)
        out["grad_elec_default"] = gl.get_grad_elec_OAO(mol, D, G)
        # a non-symmetric "RDM" pair to pin the general (unsymmetrised-input) behaviour
        rng = np.random.default_rng(seed + 500)
        Dn = rng.standard_normal((n, n))
        Gn = rng.standard_normal((n, n, n, n))
        out["nonsym_D"] = Dn
        out["nonsym_G"] = Gn
        out["nonsym_grad_elec"] = gl.get_grad_elec_OAO(mol, Dn, Gn, ao_mo_trafo=X)

        path = os.path.join(HERE, f"case_{name}.npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in out.items()})
        print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB, {len(out)} arrays")


if __name__ == "__main__":
    main()
