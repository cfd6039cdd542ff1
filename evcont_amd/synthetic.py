"""Seeded synthetic inputs of the shapes the hot path consumes.

Physical AO integrals need PySCF/libcint, which is not part of this build; the
throughput benchmark and most parity tests therefore run on random tensors with
the *symmetries* of the real objects:

* ``S_train = A A^T / T + 1`` (well conditioned), ``S_AO = B B^T / N + 1``;
* one-/two-body t-RDMs symmetric under bra<->ket exchange combined with index
  transposition and (for the two-body one) under electron-pair exchange
  (pq)<->(rs), so that all four storage layouts of the reference
  (``ab_initio_eigenvector_continuation.py:41-68``) describe the same object;
* ``eri`` 8-fold symmetric; ``hcore`` and ``dhcore`` symmetric.

Everything is float64 and generated with ``numpy.random.default_rng(seed)`` so
the same arrays can be rebuilt on any machine.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np


@dataclass
class AOArrays:
    """Array-level stand-in for the ``mol`` queries of the reference
    (``ab_initio_gradients_loewdin.py:25,147,283-284,336-339,369-370``)."""

    S: np.ndarray          # (N,N)        int1e_ovlp
    hcore: np.ndarray      # (N,N)        scf.hf.get_hcore
    eri: np.ndarray        # (N,N,N,N)    int2e, chemists' notation
    ipovlp: np.ndarray     # (3,N,N)      int1e_ipovlp
    dhcore: np.ndarray     # (A,3,N,N)    hcore_generator()(atom)
    eri_ip1: np.ndarray    # (3,N,N,N,N)  int2e_ip1
    aoslices: np.ndarray   # (A,2) int64  [start, stop) of each atom's AOs
    enuc: float
    gnuc: np.ndarray       # (A,3)
    # Declared index symmetry of ``eri`` (8-fold) and ``eri_ip1`` (last two indices), as real two-electron integrals
    # have it: True / False spare the numerical check of ``integrals_have_symmetry`` (the default compression mode
    # "auto" of the mol-level API asks once per training set), None = unknown, checked.  ``eri`` may also be handed
    # over packed, (Ms, Ms) like ``aosym="s4"``, and ``eri_ip1`` as (3,N,N,Ms) like ``aosym="s2kl"``.
    integral_symmetry: Optional[bool] = None

    @property
    def nao(self) -> int:
        return int(self.S.shape[0])

    @property
    def natm(self) -> int:
        return int(self.aoslices.shape[0])

    def pinned_packed(self) -> "AOArrays":
        """A copy whose two large arrays are packed the way PySCF delivers them with ``aosym`` -- ``eri`` as the
        (Ms, Ms) matrix of ``"s4"``, ``eri_ip1`` as (3,N,N,Ms) of ``"s2kl"`` -- and live in PINNED host memory: what an
        integral producer that writes into caller-supplied buffers hands to the MD scanner
        (``MD_utils.get_scanner``), which then uploads them without a staging copy.  Needs integrals with the
        symmetries of real ones (declared by the result)."""
        import torch
        from .hosted import staging_layout
        n, natm = self.nao, self.natm
        iu, ju = np.tril_indices(n)
        eri, ip1 = np.asarray(self.eri), np.asarray(self.eri_ip1)
        if eri.ndim == 4:
            eri = eri[iu, ju][:, iu, ju]
        if ip1.ndim == 5:
            ip1 = ip1[:, :, :, iu, ju]
        src = {"S": self.S, "hcore": self.hcore, "enuc": np.array([float(self.enuc)]), "ipovlp": self.ipovlp,
               "gnuc": self.gnuc, "eri": eri, "dhcore": self.dhcore, "eri_ip1": ip1}
        # ALL arrays in two pinned slabs laid out as the MD scanner's staging buffers are (hosted.staging_layout): the
        # scanner then uploads the slabs as they stand -- two copies per step, no host-side copy at all
        slabs, views = [], {}
        for fields, total in staging_layout(n, natm, True):
            t = torch.zeros(total, dtype=torch.float64).pin_memory()
            for k, shp, off in fields:
                v = t[off: off + int(np.prod(shp))].view(shp[1:] if k != "enuc" else shp).numpy()
                np.copyto(v, np.asarray(src[k], dtype=np.float64).reshape(v.shape))
                views[k] = v
            slabs.append(t)
        out = AOArrays(views["S"], views["hcore"], views["eri"], views["ipovlp"], views["dhcore"], views["eri_ip1"],
                       self.aoslices, float(self.enuc), views["gnuc"], integral_symmetry=True)
        out._staging_slabs = tuple(slabs)            # (keeps the pinned allocations alive)
        out._staging_key = (n, natm, True)
        return out


def equal_aoslices(nao: int, natm: int) -> np.ndarray:
    """Contiguous AO blocks, as equal as possible (first atoms get the surplus)."""
    base, extra = divmod(nao, natm)
    sizes = [base + (1 if a < extra else 0) for a in range(natm)]
    stops = np.cumsum(sizes)
    starts = stops - np.array(sizes)
    return np.stack([starts, stops], axis=1).astype(np.int64)


def make_ao_arrays(nao: int, natm: int, seed: int,
                   ao_sizes: Optional[Sequence[int]] = None,
                   degenerate_S: bool = False,
                   with_ip1: bool = True,
                   ip1_rs_symmetric: bool = False) -> AOArrays:
    """``ip1_rs_symmetric``: give ``eri_ip1`` the one index symmetry ``int2e_ip1 = (grad p q|r s)`` has,
    r <-> s (the default leaves it a general tensor, which exercises the gradient formula harder; the
    8-fold compressed t-RDM layout needs the symmetry, as real integrals have it)."""
    rng = np.random.default_rng(seed)
    n = nao
    B = rng.standard_normal((n, n))
    S = B @ B.T / n + np.eye(n)
    if degenerate_S:
        # exactly repeated overlap eigenvalues (symmetric molecules have them)
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        vals = 1.0 + 0.5 * (np.arange(n) // 2)
        S = (q * vals) @ q.T
        S = 0.5 * (S + S.T)
    h = rng.standard_normal((n, n))
    hcore = 0.5 * (h + h.T)
    e = 0.1 * rng.standard_normal((n, n, n, n))
    e = e + e.transpose(1, 0, 2, 3)
    e = e + e.transpose(0, 1, 3, 2)
    e = e + e.transpose(2, 3, 0, 1)
    eri = np.ascontiguousarray(e / 8.0)
    ipovlp = 0.1 * rng.standard_normal((3, n, n))
    dh = rng.standard_normal((natm, 3, n, n))
    dhcore = 0.5 * (dh + dh.transpose(0, 1, 3, 2))
    if with_ip1:
        eri_ip1 = 0.1 * rng.standard_normal((3, n, n, n, n))
        if ip1_rs_symmetric:
            eri_ip1 = np.ascontiguousarray(0.5 * (eri_ip1 + eri_ip1.transpose(0, 1, 2, 4, 3)))
    else:
        eri_ip1 = np.zeros((3, n, n, n, n))
    if ao_sizes is None:
        aoslices = equal_aoslices(n, natm)
    else:
        assert len(ao_sizes) == natm and sum(ao_sizes) == n
        stops = np.cumsum(ao_sizes)
        aoslices = np.stack([stops - np.array(ao_sizes), stops], axis=1).astype(np.int64)
    enuc = float(rng.standard_normal())
    gnuc = rng.standard_normal((natm, 3))
    return AOArrays(S, hcore, eri, ipovlp, dhcore, eri_ip1, aoslices, enuc, gnuc,
                    integral_symmetry=bool(ip1_rs_symmetric or not with_ip1))


def make_trdms(nao: int, ntrain: int, seed: int):
    """(S_train, one_RDM (T,T,N,N), two_RDM (T,T,N,N,N,N)) with the symmetries above."""
    rng = np.random.default_rng(seed)
    n, T = nao, ntrain
    A = rng.standard_normal((T, T))
    S_train = A @ A.T / T + np.eye(T)
    d = rng.standard_normal((T, T, n, n)) / n
    one = 0.5 * (d + d.transpose(1, 0, 3, 2))
    g = rng.standard_normal((T, T, n, n, n, n)) / (n * n)
    g = 0.5 * (g + g.transpose(0, 1, 4, 5, 2, 3))          # (pq)<->(rs)
    g = 0.5 * (g + g.transpose(1, 0, 3, 2, 5, 4))          # bra<->ket with p<->q, r<->s
    return S_train, np.ascontiguousarray(one), np.ascontiguousarray(g)


def pack_rows(two_rdm_full: np.ndarray, pair_sym: bool, elec_sym: bool) -> np.ndarray:
    """Re-express a (T,T,N,N,N,N) t-RDM in one of the reference's other layouts
    (``ab_initio_eigenvector_continuation.py:45-68``): pairs (a>=b) in
    ``np.tril_indices`` order, columns as the row-major lower triangle of the
    (N^2,N^2) matrix (``electron_integral_utils.py:57``)."""
    T = two_rdm_full.shape[0]
    n = two_rdm_full.shape[2]
    g = two_rdm_full
    if elec_sym:
        r, c = np.tril_indices(n * n)
        g = g.reshape(T, T, n * n, n * n)[:, :, r, c]
    if pair_sym:
        a, b = np.tril_indices(T)
        g = g[a, b]
    return np.ascontiguousarray(g)


# --------------------------------------------------------------------------
# Device-side generators for the throughput benchmark: same
# symmetries, generated with torch.Generator(device) so that multi-GB inputs
# never cross PCIe.  Imported lazily so that the numpy part works without torch.
# --------------------------------------------------------------------------
def make_device_ao(nao: int, natm: int, seed: int, device, ao_sizes: Optional[Sequence[int]] = None,
                   ip1_rs_symmetric: bool = False):
    """One synthetic geometry resident on the device (evaluator.DeviceAO)."""
    import torch
    from .evaluator import DeviceAO
    g = torch.Generator(device=device).manual_seed(int(seed))
    n = nao
    rn = lambda *s: torch.randn(*s, generator=g, device=device, dtype=torch.float64)
    B = rn(n, n)
    S = B @ B.T / n + torch.eye(n, device=device, dtype=torch.float64)
    h = rn(n, n)
    hcore = 0.5 * (h + h.T)
    e = 0.1 * rn(n, n, n, n)
    e = e + e.permute(1, 0, 2, 3)
    e = e + e.permute(0, 1, 3, 2)
    e = (e + e.permute(2, 3, 0, 1)) / 8.0
    ipovlp = 0.1 * rn(3, n, n)
    dh = rn(natm, 3, n, n)
    dhcore = 0.5 * (dh + dh.transpose(2, 3))
    ip1 = 0.1 * rn(3, n, n, n, n)
    if ip1_rs_symmetric:
        ip1 = 0.5 * (ip1 + ip1.transpose(3, 4))
    if ao_sizes is None:
        sl = equal_aoslices(n, natm)
    else:
        stops = np.cumsum(ao_sizes)
        sl = np.stack([stops - np.array(ao_sizes), stops], axis=1).astype(np.int64)
    enuc = float(rn(1).item())
    gnuc = rn(natm, 3)
    return DeviceAO(S=S.contiguous(), hcore=hcore.contiguous(), eri=e.contiguous(), enuc=enuc, natm=natm,
                    ipovlp=ipovlp.contiguous(), dhcore=dhcore.contiguous(), eri_ip1=ip1.contiguous(),
                    gnuc=gnuc.contiguous(), aoslices=torch.from_numpy(sl).to(device))


def make_device_trdm_rows(nao: int, ntrain: int, layout: int, seed: int, device, row_range=None):
    """(S_train, one_RDM (T,T,N,N), two-body rows [r0:r1) as a (r1-r0, cols) tensor) on the device.

    Row r of the two-body matrix is generated from its own seed, so a rank that owns a slice
    of the rows builds exactly the rows the single-device run would hold."""
    import torch
    from .evaluator import layout_shape
    n, T = nao, ntrain
    g = torch.Generator(device=device).manual_seed(int(seed))
    A = torch.randn(T, T, generator=g, device=device, dtype=torch.float64)
    S_train = A @ A.T / T + torch.eye(T, device=device, dtype=torch.float64)
    d = torch.randn(T, T, n, n, generator=g, device=device, dtype=torch.float64) / n
    one = 0.5 * (d + d.permute(1, 0, 3, 2))
    rows, cols = layout_shape(layout, T, n)
    r0, r1 = row_range if row_range is not None else (0, rows)
    two = torch.empty((r1 - r0, cols), dtype=torch.float64, device=device)
    gr = torch.Generator(device=device)
    for r in range(r0, r1):
        gr.manual_seed(int(seed) * 100003 + r)
        torch.randn(cols, generator=gr, device=device, dtype=torch.float64, out=two[r - r0])
    two.mul_(1.0 / (n * n))
    return S_train.contiguous(), one.contiguous(), two
