"""TEST INFRASTRUCTURE: a stand-in for the part of PySCF's surface the hot path touches, so that the
``pyscf`` branches of ``evcont_amd`` (``integrals.ao_arrays(Mole)``, ``MD_utils.get_scanner`` as a
``lib.GradScanner``, ``MD_utils.get_trajectory`` through ``pyscf.md.NVE``, integrator callbacks reading
``locals["scanner"].base``) execute in an image that has no PySCF.

What is imitated is PySCF's CALL PATTERN, not its arithmetic:

* ``StubMole`` answers the queries the reference makes of a ``gto.Mole``
  (``ab_initio_gradients_loewdin.py:25,147,283-284,336-339,369-370``; ``MD_utils.py:20-57``):
  ``intor("int1e_ovlp" | "int1e_ipovlp" | "int2e" | "int2e_ip1", comp=, aosym=)``, ``nao``, ``natm``,
  ``aoslice_by_atom()``, ``energy_nuc()``, ``atom_coords()``, ``atom_mass_list()``, ``set_geom_()``, ``build()``,
  ``copy()``, ``atom``, ``stdout``, ``verbose``, ``incore_anyway`` -- from a fixed set of arrays (golden fixtures) or
  from the closed-form hydrogen-chain integrals of ``evcont_amd.hchain`` (then it can move).
* ``md.NVE`` / ``md.integrators.NVTBerendson`` follow the control flow of ``pyscf/md/integrators.py``: the
  integrator takes a ``lib.GradScanner`` instance, reads ``scanner.mol``, calls ``scanner(mol)`` ONCE per step after
  ``mol.set_geom_(...)``, checks ``scanner.converged``, appends a frame with ``.coord/.veloc/.ekin/.epot/.time`` to
  ``frames`` when ``incore_anyway``, and calls ``callback(locals())`` with ``mol`` and ``scanner`` among the keys.

``installed()`` is a context manager that puts the stand-in into ``sys.modules`` and removes it afterwards.
"""
from __future__ import annotations

import contextlib
import sys
import types

import numpy as np

AMU2AU = 1822.888486209


class StubMole:
    def __init__(self, arrays=None, coords=None, factory=None):
        """``arrays``: an object with the AOArrays fields (static molecule); or ``coords`` + ``factory(coords)``
        returning such an object (a molecule that can be rebuilt at new coordinates)."""
        self._factory = factory
        self._coords = None if coords is None else np.array(coords, dtype=np.float64)
        self._ao = arrays if arrays is not None else factory(self._coords)
        self.stdout = sys.stdout
        self.verbose = 0
        self.incore_anyway = False
        self.queries = []          # names of the intor calls made (tests look at them)
        # A real Mole serves libcint integrals, which have the index symmetries of real two-electron integrals; this
        # stand-in may be built from seeded general tensors (the golden fixtures' eri_ip1): it says so, and the
        # default compression mode of evcont_amd then keeps the caller's layout for it
        self.integral_symmetry = getattr(self._ao, "integral_symmetry", None)
        if self.integral_symmetry is None:
            from evcont_amd.ab_initio_eigenvector_continuation import integrals_have_symmetry
            self.integral_symmetry = bool(integrals_have_symmetry(self._ao))

    # -- what the reference asks a Mole --------------------------------------------------
    @property
    def nao(self):
        return int(self._ao.S.shape[0])

    @property
    def natm(self):
        return int(np.asarray(self._ao.aoslices).shape[0])

    @property
    def atom(self):
        return [("H", tuple(x)) for x in self.atom_coords()]

    def intor(self, name, comp=None, aosym="s1", out=None):
        res = self._intor(name, aosym)
        self.queries.append((name, aosym, out is not None))
        if out is not None:
            np.copyto(out, res.reshape(out.shape))
            return out
        return res

    def _intor(self, name, aosym):
        n = self.nao
        if name == "int1e_ovlp":
            return np.array(self._ao.S)
        if name == "int1e_ipovlp":
            return np.array(self._ao.ipovlp)
        if name == "int2e":
            eri = np.asarray(self._ao.eri).reshape(n, n, n, n)
            if aosym == "s4":
                iu, ju = np.tril_indices(n)
                return np.ascontiguousarray(eri[iu, ju][:, iu, ju])
            return np.array(eri)
        if name == "int2e_ip1":
            ip1 = np.asarray(self._ao.eri_ip1).reshape(3, n, n, n, n)
            if aosym == "s2kl":
                iu, ju = np.tril_indices(n)
                return np.ascontiguousarray(ip1[:, :, :, iu, ju])
            return np.array(ip1)
        raise KeyError(name)

    def aoslice_by_atom(self):
        return [(0, 0, int(a), int(b)) for a, b in np.asarray(self._ao.aoslices)]

    def energy_nuc(self):
        return float(self._ao.enuc)

    def atom_coords(self):
        if self._coords is None:
            return np.zeros((self.natm, 3))
        return np.array(self._coords)

    def atom_mass_list(self, isotope_avg=False):
        return np.full(self.natm, 1.008)

    def set_geom_(self, coords, unit="B"):
        assert self._factory is not None, "this StubMole serves fixed arrays"
        self._coords = np.array(coords, dtype=np.float64)
        self._ao = self._factory(self._coords)
        return self

    def build(self, *a, **k):
        return self

    def copy(self):
        m = StubMole(self._ao, self._coords, self._factory)
        return m


def _make_modules():
    ps = types.ModuleType("pyscf")
    sub = {n: types.ModuleType("pyscf." + n) for n in ("scf", "grad", "lib", "md", "gto")}
    for n, m in sub.items():
        setattr(ps, n, m)

    class GradScanner:                                  # pyscf.lib.GradScanner: `converged` forwards to base
        def __init__(self, g=None):
            self.base = getattr(g, "base", None)

        @property
        def converged(self):
            return self.base.converged

    sub["lib"].GradScanner = GradScanner
    sub["scf"].hf = types.SimpleNamespace(get_hcore=lambda mol: np.array(mol._ao.hcore))
    sub["scf"].RHF = lambda mol: types.SimpleNamespace(mol=mol)

    class _Grad:                                        # grad.RHF(scf.RHF(mol))
        def __init__(self, mf):
            self.mol = mf.mol

        def grad_nuc(self):
            return np.array(self.mol._ao.gnuc)

        def hcore_generator(self):
            return lambda ia: np.array(self.mol._ao.dhcore[ia])

    sub["grad"].RHF = _Grad

    class Frame:
        def __init__(self, integ):
            self.ekin, self.epot, self.etot = integ.ekin, integ.epot, integ.ekin + integ.epot
            self.coord, self.veloc, self.time = integ.mol.atom_coords(), np.array(integ.veloc), integ.time

    class _Integrator:
        """Control flow of pyscf/md/integrators.py::_Integrator + VelocityVerlet."""

        def __init__(self, method, **kwargs):
            assert isinstance(method, GradScanner), "PySCF accepts a GradScanner (or a method it can scan)"
            self.scanner = method
            self.mol = self.scanner.mol
            self.incore_anyway = self.mol.incore_anyway
            self.veloc = None
            self.steps, self.dt = 1, 10
            self.frames = None
            self.epot = self.ekin = None
            self.time = 0
            self.data_output = self.trajectory_output = self.energy_output = self.callback = None
            self.accel = None
            self.__dict__.update(kwargs)
            self._masses = np.asarray(self.mol.atom_mass_list()) * AMU2AU

        def compute_kinetic_energy(self):
            return 0.5 * float(np.sum(self._masses[:, None] * self.veloc ** 2))

        def _compute_accel(self):
            e_tot, grad = self.scanner(self.mol)
            if not self.scanner.converged:
                raise RuntimeError("Gradients did not converge!")
            return e_tot, -1.0 * np.asarray(grad) / self._masses.reshape(-1, 1)

        def _scale_velocities(self):
            pass

        def _next(self):
            if self.accel is None:
                next_epot, next_accel = self._compute_accel()
            else:
                R = self.mol.atom_coords() + self.dt * self.veloc + 0.5 * self.dt ** 2 * self.accel
                self.mol.set_geom_(R, unit="B")
                self.mol.build()
                next_epot, next_accel = self._compute_accel()
                self.veloc = self.veloc + 0.5 * self.dt * (self.accel + next_accel)
                self._scale_velocities()
                self.time += self.dt
            self.epot, self.accel = next_epot, next_accel
            self.ekin = self.compute_kinetic_energy()
            return Frame(self)

        def run(self, veloc=None, steps=None):
            if veloc is not None:
                self.veloc = veloc
            if steps is not None:
                self.steps = steps
            if self.veloc is None:
                self.veloc = np.zeros((self.mol.natm, 3))
            if self.frames is None and self.incore_anyway:
                self.frames = []
            for iteration in range(self.steps):
                frame = self._next()
                if self.incore_anyway:
                    self.frames.append(frame)
                if callable(self.callback):
                    mol = self.mol                      # noqa: F841 (what PySCF exposes through locals())
                    scanner = self.scanner              # noqa: F841
                    integrator = self                   # noqa: F841
                    self.callback(locals())
            return self

        kernel = run

    class NVE(_Integrator):
        pass

    class NVTBerendson(_Integrator):
        def __init__(self, method, T, taut, **kwargs):
            self.T, self.taut = T, taut
            super().__init__(method, **kwargs)

        def _scale_velocities(self):
            kB = 3.166811563e-6                         # Hartree / K
            ndof = 3 * self.mol.natm
            Tnow = 2.0 * self.compute_kinetic_energy() / (ndof * kB)
            if Tnow > 0:
                self.veloc = self.veloc * np.sqrt(1.0 + (self.T / Tnow - 1.0) * self.dt / self.taut)

    sub["md"].NVE = NVE
    sub["md"].integrators = types.SimpleNamespace(NVE=NVE, NVTBerendson=NVTBerendson, VelocityVerlet=NVE)
    mods = {"pyscf": ps}
    mods.update({"pyscf." + n: m for n, m in sub.items()})
    return mods


@contextlib.contextmanager
def installed():
    """``with pyscf_stub.installed(): ...`` -- the stand-in is importable as ``pyscf`` inside the block."""
    assert "pyscf" not in sys.modules or getattr(sys.modules["pyscf"], "_evc_stub", False), \
        "a real PySCF is importable here: use it instead of the stand-in"
    mods = _make_modules()
    mods["pyscf"]._evc_stub = True
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        yield mods["pyscf"]
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
