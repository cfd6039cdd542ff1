"""Device-resident continuation evaluator.

``DeviceTRDMs`` uploads the training data (overlap, one- and two-body transition RDMs,
``FCI_EVCont.py:106-131``) ONCE, in whichever of the reference's four layouts the caller
holds (``ab_initio_eigenvector_continuation.py:41-68``), as a ``(rows, ld)`` float64
matrix in HBM with 128-byte-aligned rows.  ``DeviceAO`` holds the AO integrals of one
geometry.  ``ContinuationEvaluator`` owns the workspace and the output buffers and
enqueues the per-geometry DAG through the C ABI (``include/evcont_hip.h``); results stay
on the device until the caller reads them.

PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import weakref

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import TrdmSet, Geometry, Outputs, check

F64 = torch.float64

# EVCONT_AMD_CHECK_SYM: the compressed layout (EVC_LAYOUT_SYM8) and the packed s4 / s2kl inputs are exact only for AO
# integrals with the index symmetries of real two-electron integrals (include/evcont_hip.h).  "1" (default): every
# evaluator on the compressed layout verifies them on its FIRST call (and every host-side packing helper on its first
# use) and raises instead of returning wrong forces; "2": on every call; "0": never.
_CHECK_SYM = os.environ.get("EVCONT_AMD_CHECK_SYM", "1")
_host_checks_done = set()


def check_integral_symmetry(eri, eri_ip1, n: int, tol: float = 1.0e-9, what: str = "") -> None:
    """Raise ``EvcontHipError`` unless ``eri`` is 8-fold symmetric ((pq|rs) = (qp|rs) = (pq|sr) = (rs|pq)) and
    ``eri_ip1[x,p,q,r,s] = eri_ip1[x,p,q,s,r]``, to ``tol`` relative to the largest element.  Arrays may be numpy or
    torch, with any number of leading batch axes; ``eri`` may be the packed s4 matrix (Ms, Ms) (then only the
    pair-exchange symmetry is left to check), ``eri_ip1`` packed s2kl (nothing left to check) or None."""
    npr = n * (n + 1) // 2
    is_t = torch.is_tensor(eri)
    mx = (lambda x: float(x.abs().max().item())) if is_t else (lambda x: float(np.abs(x).max()))
    bad = []
    if eri is not None and n > 1:
        if eri.shape[-1] == npr and eri.shape[-2] == npr and (eri.ndim < 4 or eri.shape[-3] != n):
            e = eri.reshape(-1, npr, npr)
            sw = (lambda x, a, b: x.transpose(a, b)) if is_t else (lambda x, a, b: np.swapaxes(x, a, b))
            scale = max(mx(e), 1e-300)
            if mx(e - sw(e, 1, 2)) > tol * scale:
                bad.append("eri (packed s4): (pq|rs) != (rs|pq)")
        else:
            e = eri.reshape(-1, n, n, n, n)
            sw = (lambda x, a, b: x.transpose(a, b)) if is_t else (lambda x, a, b: np.swapaxes(x, a, b))
            scale = max(mx(e), 1e-300)
            if mx(e - sw(e, 1, 2)) > tol * scale:
                bad.append("eri: (pq|rs) != (qp|rs)")
            if mx(e - sw(e, 3, 4)) > tol * scale:
                bad.append("eri: (pq|rs) != (pq|sr)")
            if mx(e - sw(sw(e, 1, 3), 2, 4)) > tol * scale:
                bad.append("eri: (pq|rs) != (rs|pq)")
    if eri_ip1 is not None and n > 1 and eri_ip1.shape[-1] == n and eri_ip1.ndim >= 5:
        x = eri_ip1.reshape(-1, n, n, n, n)
        sw = (lambda y, a, b: y.transpose(a, b)) if torch.is_tensor(x) else (lambda y, a, b: np.swapaxes(y, a, b))
        mxx = (lambda y: float(y.abs().max().item())) if torch.is_tensor(x) else (lambda y: float(np.abs(y).max()))
        scale = max(mxx(x), 1e-300)
        if mxx(x - sw(x, 3, 4)) > tol * scale:
            bad.append("eri_ip1[x,p,q,r,s] != eri_ip1[x,p,q,s,r]")
    if bad:
        raise _lib.EvcontHipError(
            "AO integrals without the index symmetries of real two-electron integrals (" + "; ".join(bad) + ")"
            + (f" in {what}" if what else "") + ": the 8-fold compressed t-RDM layout (compress='sym8') and the packed "
            "s4 / s2kl inputs would give wrong energies / forces for them.  Use the reference layouts (compress=None) "
            "for such tensors; EVCONT_AMD_CHECK_SYM=0 disables this check.")


def spot_check_integral_symmetry(eri, eri_ip1, n: int, samples: int = 4096, tol: float = 1.0e-9, what: str = "") -> None:
    """The same test on ``samples`` random index quadruples of full host arrays (numpy): what the host-side packing
    helpers run on EVERY call after their first, complete check -- packing keeps one triangle, so a later geometry (or
    another integral source) without the symmetries could not be noticed afterwards (EVCONT_AMD_CHECK_SYM=0: never)."""
    if n < 2:
        return
    rng = np.random.default_rng()
    p, q, r, t = (rng.integers(0, n, samples) for _ in range(4))
    bad = []
    e = np.asarray(eri)
    if e.ndim >= 4 and e.shape[-1] == n and e.shape[-4] == n:
        e = e.reshape(-1, n, n, n, n)[0]
        v = e[p, q, r, t]
        scale = max(float(np.abs(v).max()), 1e-300)
        if max(np.abs(v - e[q, p, r, t]).max(), np.abs(v - e[p, q, t, r]).max(), np.abs(v - e[r, t, p, q]).max()) > tol * scale:
            bad.append("eri is not 8-fold symmetric")
    if eri_ip1 is not None:
        x = np.asarray(eri_ip1)
        if x.ndim >= 5 and x.shape[-1] == n and x.shape[-2] == n:
            x = x.reshape(-1, n, n, n, n)
            c = rng.integers(0, x.shape[0], samples)
            v = x[c, p, q, r, t]
            scale = max(float(np.abs(v).max()), 1e-300)
            if np.abs(v - x[c, p, q, t, r]).max() > tol * scale:
                bad.append("eri_ip1[x,p,q,r,s] != eri_ip1[x,p,q,s,r]")
    if bad:
        raise _lib.EvcontHipError(
            "AO integrals without the index symmetries of real two-electron integrals (" + "; ".join(bad) + ")"
            + (f" in {what}" if what else "") + ": packing them (s4 / s2kl) would give wrong energies / forces.  Use the "
            "reference layouts (compress=None) for such tensors; EVCONT_AMD_CHECK_SYM=0 disables this check.")


def _host_check_once(tag: str) -> bool:
    """Whether a host-side packing helper should verify the symmetries now (EVCONT_AMD_CHECK_SYM)."""
    if _CHECK_SYM == "0":
        return False
    if _CHECK_SYM == "2" or tag not in _host_checks_done:
        _host_checks_done.add(tag)
        return True
    return False


def _dev(device=None) -> torch.device:
    if device is None:
        if not torch.cuda.is_available():
            raise _lib.EvcontHipError("no HIP device visible: evcont_amd needs an MI355X (there is no CPU fallback)")
        return torch.device("cuda", torch.cuda.current_device())
    d = torch.device(device)
    if d.type != "cuda":
        raise _lib.EvcontHipError(f"device {d} is not a HIP device")
    return d


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def layout_shape(layout: int, T: int, n: int) -> Tuple[int, int]:
    """(rows, cols) of the matrix view of a two-body t-RDM in the given layout."""
    n2 = n * n
    rows = T * (T + 1) // 2 if layout in (5, 2, _lib.LAYOUT_SYM8) else T * T
    if layout == _lib.LAYOUT_SYM8:
        ms = n * (n + 1) // 2
        return rows, ms * (ms + 1) // 2
    cols = n2 * (n2 + 1) // 2 if layout in (3, 2) else n2 * n2
    return rows, cols


def infer_layout(two_RDM, T: int, n: int) -> int:
    nd = two_RDM.ndim
    assert nd in (6, 5, 3, 2), "two_RDM must have 6, 5, 3 or 2 dimensions"  # evcont.py:70-71
    rows, cols = layout_shape(nd, T, n)
    got = (int(np.prod(two_RDM.shape[: 2 if nd in (6, 3) else 1])), int(np.prod(two_RDM.shape[2 if nd in (6, 3) else 1:])))
    if got != (rows, cols):
        raise ValueError(f"two_RDM shape {tuple(two_RDM.shape)} inconsistent with T={T}, N={n} (layout ndim {nd})")
    return nd


def sym8_column_images(layout: int, n: int):
    """For every column ``u(u+1)/2+v`` of the 8-fold compressed layout (``u = i(i+1)/2+j``, ``i >= j``, ``v`` likewise
    from ``k >= l``, ``u >= v``) the columns of a SOURCE layout (ndim 6/5: ``N^4`` unpacked, 3/2: lower triangle of the
    ``(N^2,N^2)`` matrix) that hold the eight images of ``(i,j,k,l)`` under the index permutations of real
    two-electron integrals: a list of eight int64 arrays.  The compressed entry is the mean over them."""
    iu, ju = np.tril_indices(n)
    U, V = np.tril_indices(len(iu))
    i, j, k, l = iu[U], ju[U], iu[V], ju[V]
    if layout in (3, 2):
        def col(a_, b_, c_, d_):
            R, Cc = a_ * n + b_, c_ * n + d_
            hi, lo = np.maximum(R, Cc).astype(np.int64), np.minimum(R, Cc).astype(np.int64)
            return hi * (hi + 1) // 2 + lo
    else:
        def col(a_, b_, c_, d_):
            return ((a_.astype(np.int64) * n + b_) * n + c_) * n + d_
    images = [(i, j, k, l), (j, i, k, l), (i, j, l, k), (j, i, l, k),
              (k, l, i, j), (l, k, i, j), (k, l, j, i), (l, k, j, i)]
    return [np.ascontiguousarray(col(*im)) for im in images]


def _upload_rows(src: np.ndarray, rows: int, cols: int, r0: int, r1: int, device) -> Tuple[torch.Tensor, int]:
    """Copy rows [r0,r1) of the (rows, cols) view of `src` into a zero-padded (r1-r0, ld) device matrix."""
    ld = (cols + 15) // 16 * 16
    out = torch.zeros((max(r1 - r0, 1), ld), dtype=F64, device=device)
    lead = src.shape[: src.ndim - (4 if src.ndim in (6, 5) else 1)]
    step = max(1, (256 << 20) // (cols * 8))
    for a in range(r0, r1, step):
        b = min(r1, a + step)
        idx = np.unravel_index(np.arange(a, b), lead)
        block = np.ascontiguousarray(src[idx], dtype=np.float64).reshape(b - a, cols)
        out[a - r0:b - r0, :cols].copy_(torch.from_numpy(block), non_blocking=False)
    return out, ld


def _pad_even(mat: torch.Tensor) -> torch.Tensor:
    """(rows, cols) -> contiguous (rows, ld) with ld even (16-byte aligned rows), zero padded."""
    rows, cols = mat.shape
    if cols % 2 == 0:
        return mat.contiguous()
    out = torch.zeros((rows, cols + 1), dtype=F64, device=mat.device)
    out[:, :cols].copy_(mat)
    return out


class DeviceTRDMs:
    """Training data resident in HBM.  ``row_range`` selects the slice of two-body rows this
    rank owns (pair sharding); the one-body t-RDM and S are replicated."""

    def __init__(self, one_RDM, two_RDM, S, device=None, row_range: Optional[Tuple[int, int]] = None,
                 compress: Optional[str] = None):
        """``compress="sym8"``: keep only the 8-fold symmetrised part of the two-body t-RDMs
        (``compress_sym8_``, include/evcont_hip.h ``EVC_LAYOUT_SYM8``)."""
        self.device = _dev(device)
        one_RDM = np.asarray(one_RDM) if not torch.is_tensor(one_RDM) else one_RDM
        S = np.asarray(S) if not torch.is_tensor(S) else S
        T = int(S.shape[0])
        n = int(one_RDM.shape[-1])
        assert tuple(one_RDM.shape) == (T, T, n, n), "one_RDM must be (T,T,N,N)"
        self.T, self.n = T, n
        if torch.is_tensor(two_RDM):
            self.layout = infer_layout(two_RDM, T, n)
            rows, cols = layout_shape(self.layout, T, n)
            r0, r1 = row_range if row_range is not None else (0, rows)
            mat = two_RDM.reshape(rows, cols)[r0:r1].to(self.device, F64)
            ld = (cols + 15) // 16 * 16
            self.two = torch.zeros((max(r1 - r0, 1), ld), dtype=F64, device=self.device)
            self.two[: r1 - r0, :cols].copy_(mat)
        else:
            two_RDM = np.asarray(two_RDM)
            self.layout = infer_layout(two_RDM, T, n)
            rows, cols = layout_shape(self.layout, T, n)
            r0, r1 = row_range if row_range is not None else (0, rows)
            self.two, ld = _upload_rows(two_RDM, rows, cols, r0, r1, self.device)
        assert 0 <= r0 <= r1 <= rows
        self.rows_total, self.cols, self.ld = rows, cols, ld
        self.row_offset, self.rows_local = r0, r1 - r0
        as_t = lambda x: (x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)))
        self.one = _pad_even(as_t(one_RDM).to(self.device, F64).reshape(T * T, n * n))
        self.S = as_t(S).to(self.device, F64).contiguous()
        self.cstruct = TrdmSet(n=n, ntrain=T, layout=self.layout, reserved=0, rows2=self.rows_local,
                               row_offset=self.row_offset, rows2_total=rows, cols2=cols, ld2=ld,
                               ld1=self.one.shape[1],
                               two_rdm=self.two.data_ptr(), one_rdm=self.one.data_ptr(),
                               s_train=self.S.data_ptr())
        if compress is not None:
            if compress != "sym8":
                raise ValueError(f"unknown t-RDM compression {compress!r} (known: 'sym8')")
            self.compress_sym8_()

    def compress_sym8_(self) -> "DeviceTRDMs":
        """Replace the resident two-body t-RDMs by their 8-fold compressed form (``EVC_LAYOUT_SYM8``,
        in place): per pair ``a >= b`` the mean of ``Gamma`` over the index permutations under which real
        two-electron integrals are invariant, stored once per class -- N(N+1)/2 (N(N+1)/2 + 1)/2 columns
        instead of N^2 (N^2+1)/2 (3.7x fewer bytes at N = 30).

        Exact (to rounding) for the Hermitian continuation whenever the AO integrals carry those symmetries
        (``eri`` 8-fold, ``eri_ip1`` symmetric in its last two indices: every PySCF ``int2e`` /
        ``int2e_ip1``); ``predicted_two_rdm`` then is the symmetrised 2-RDM.  Not for ``hermitian=False``."""
        if self.layout == _lib.LAYOUT_SYM8:
            return self
        T, n, d = self.T, self.n, self.device
        if self.layout in (6, 3):
            if (self.row_offset, self.rows_local) != (0, self.rows_total):
                raise ValueError("compress_sym8_: shard the pair layouts (5, 2), not the (T,T,...) ones")
            a, b = np.tril_indices(T)
            src = torch.from_numpy((a * T + b).astype(np.int64)).to(d)
            r0 = 0
        else:
            src = torch.arange(self.rows_local, dtype=torch.int64, device=d)
            r0 = self.row_offset
        idx = [torch.from_numpy(ix).to(d) for ix in sym8_column_images(self.layout, n)]
        rows8, cols8 = layout_shape(_lib.LAYOUT_SYM8, T, n)
        ld8 = (cols8 + 15) // 16 * 16
        nloc = int(src.numel())
        out = torch.zeros((max(nloc, 1), ld8), dtype=F64, device=d)
        step = max(1, (256 << 20) // (self.ld * 8))
        for a0 in range(0, nloc, step):
            blk = self.two.index_select(0, src[a0:a0 + step])
            acc = blk.index_select(1, idx[0])
            for ix in idx[1:]:
                acc += blk.index_select(1, ix)
            out[a0:a0 + acc.shape[0], :cols8] = acc * 0.125
            del blk, acc
        self.two, self.layout = out, _lib.LAYOUT_SYM8
        self.rows_total, self.cols, self.ld = rows8, cols8, ld8
        self.row_offset, self.rows_local = r0, nloc
        self.cstruct = TrdmSet(n=n, ntrain=T, layout=self.layout, reserved=0, rows2=self.rows_local,
                               row_offset=self.row_offset, rows2_total=rows8, cols2=cols8, ld2=ld8,
                               ld1=self.one.shape[1],
                               two_rdm=self.two.data_ptr(), one_rdm=self.one.data_ptr(),
                               s_train=self.S.data_ptr())
        return self

    @classmethod
    def from_device_rows(cls, one_RDM: torch.Tensor, two_rows: torch.Tensor, S: torch.Tensor, layout: int,
                         row_offset: int = 0, rows_total: Optional[int] = None) -> "DeviceTRDMs":
        """Adopt two-body rows already on the device: ``two_rows`` is the (rows_local, cols) slice
        [row_offset, row_offset+rows_local) of the layout's matrix view (no host round trip)."""
        self = cls.__new__(cls)
        self.device = two_rows.device
        T, n = int(S.shape[0]), int(one_RDM.shape[-1])
        self.T, self.n, self.layout = T, n, int(layout)
        rows, cols = layout_shape(self.layout, T, n)
        if rows_total is not None:
            assert rows_total == rows
        assert two_rows.ndim == 2 and two_rows.shape[1] == cols and two_rows.dtype == F64
        r0, r1 = int(row_offset), int(row_offset) + int(two_rows.shape[0])
        assert 0 <= r0 <= r1 <= rows
        ld = (cols + 15) // 16 * 16
        if ld == cols and two_rows.is_contiguous() and two_rows.shape[0] > 0:
            self.two = two_rows
        else:
            self.two = torch.zeros((max(r1 - r0, 1), ld), dtype=F64, device=self.device)
            self.two[: r1 - r0, :cols].copy_(two_rows)
        self.rows_total, self.cols, self.ld = rows, cols, ld
        self.row_offset, self.rows_local = r0, r1 - r0
        self.one = _pad_even(one_RDM.to(self.device, F64).reshape(T * T, n * n))
        self.S = S.to(self.device, F64).contiguous()
        self.cstruct = TrdmSet(n=n, ntrain=T, layout=self.layout, reserved=0, rows2=self.rows_local,
                               row_offset=self.row_offset, rows2_total=rows, cols2=cols, ld2=ld,
                               ld1=self.one.shape[1],
                               two_rdm=self.two.data_ptr(), one_rdm=self.one.data_ptr(),
                               s_train=self.S.data_ptr())
        return self

    @property
    def nbytes_streamed_per_pass(self) -> int:
        """Algorithmic bytes one pass over the local two-body rows reads."""
        return self.rows_local * self.cols * 8


@dataclass
class DeviceAO:
    """AO integrals of one geometry on the device (fields as in include/evcont_hip.h evc_geometry)."""
    S: torch.Tensor
    hcore: torch.Tensor
    eri: torch.Tensor
    enuc: float
    natm: int = 0
    ipovlp: Optional[torch.Tensor] = None
    dhcore: Optional[torch.Tensor] = None
    eri_ip1: Optional[torch.Tensor] = None
    gnuc: Optional[torch.Tensor] = None
    aoslices: Optional[torch.Tensor] = None
    ip1_s2kl: bool = False   # eri_ip1 is (3,N,N,N(N+1)/2): packed in its last two indices (EVC_FLAG_IP1_S2KL)
    eri_s4: bool = False     # eri is (N(N+1)/2, N(N+1)/2): packed in both index pairs (EVC_FLAG_ERI_S4)

    @property
    def nao(self) -> int:
        return int(self.S.shape[0])

    @staticmethod
    def from_arrays(ao, device=None, energy_only: bool = False, pack_ip1: bool = False,
                    pack_eri: bool = False) -> "DeviceAO":
        """From any object with the AOArrays fields (numpy).  ``pack_ip1``: upload ``eri_ip1`` packed in its last
        two AO indices (half the bytes; what ``mol.intor("int2e_ip1", aosym="s2kl")`` returns -- an ``eri_ip1`` that
        already has that shape is taken as is); ``pack_eri``: upload ``eri`` packed in both index pairs (a quarter of
        the bytes, ``aosym="s4"``; a 2-D ``eri`` of that shape is taken as is).  Both only for evaluators on the
        compressed ``sym8`` layout."""
        d = _dev(device)
        up = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(d)
        n = int(np.asarray(ao.S).shape[0])
        npr = n * (n + 1) // 2
        eri = np.asarray(ao.eri)
        s4 = eri.ndim == 2 and eri.shape == (npr, npr) and n > 1
        if (pack_eri or pack_ip1) and not energy_only:
            # complete check at the first use (per orbital count), a random sample on every later call
            if _host_check_once(f"DeviceAO.from_arrays/{n}"):
                check_integral_symmetry(eri, np.asarray(ao.eri_ip1), n, what="DeviceAO.from_arrays(pack_...=True)")
            elif _CHECK_SYM != "0":
                spot_check_integral_symmetry(eri, ao.eri_ip1, n, what="DeviceAO.from_arrays(pack_...=True)")
        if pack_eri and not s4:
            iu, ju = np.tril_indices(n)
            eri = eri.reshape(n, n, n, n)[iu, ju][:, iu, ju]
            s4 = True
        out = DeviceAO(S=up(ao.S), hcore=up(ao.hcore), eri=up(eri), enuc=float(ao.enuc),
                       natm=int(np.asarray(ao.aoslices).shape[0]), eri_s4=s4)
        if not energy_only:
            ip1 = np.asarray(ao.eri_ip1)
            n = out.nao
            if ip1.ndim == 4 or (ip1.ndim == 3 and ip1.shape[-1] == n * (n + 1) // 2):   # already s2kl
                ip1 = ip1.reshape(3, n, n, n * (n + 1) // 2)
                out.ip1_s2kl = True
            elif pack_ip1:
                iu, ju = np.tril_indices(n)
                ip1 = ip1.reshape(3, n, n, n, n)[:, :, :, iu, ju]
                out.ip1_s2kl = True
            out.ipovlp, out.dhcore, out.eri_ip1, out.gnuc = up(ao.ipovlp), up(ao.dhcore), up(ip1), up(ao.gnuc)
            out.aoslices = torch.from_numpy(np.ascontiguousarray(ao.aoslices, dtype=np.int64)).to(d)
        return out

    def packed_ip1(self, eri: bool = False) -> "DeviceAO":
        """A copy (sharing every other array) whose ``eri_ip1`` is packed in its last two indices (and, with
        ``eri=True``, whose ``eri`` is packed in both index pairs), gathered on the device from the full arrays."""
        n = self.nao
        iu, ju = np.tril_indices(n)
        idx = torch.from_numpy((iu * n + ju).astype(np.int64)).to(self.S.device)
        ip1, s2kl = self.eri_ip1, self.ip1_s2kl
        if ip1 is not None and not s2kl:
            ip1, s2kl = ip1.reshape(3, n, n, n * n).index_select(3, idx).contiguous(), True
        e, s4 = self.eri, self.eri_s4
        if eri and not s4:
            e, s4 = e.reshape(n * n, n * n).index_select(0, idx).index_select(1, idx).contiguous(), True
        return DeviceAO(S=self.S, hcore=self.hcore, eri=e, enuc=self.enuc, natm=self.natm, ipovlp=self.ipovlp,
                        dhcore=self.dhcore, eri_ip1=ip1, gnuc=self.gnuc, aoslices=self.aoslices, ip1_s2kl=s2kl,
                        eri_s4=s4)

    def cstruct(self) -> Geometry:
        p = lambda t: (t.data_ptr() if t is not None else None)
        return Geometry(natm=self.natm, reserved=0, enuc=self.enuc, S=p(self.S), hcore=p(self.hcore),
                        eri=p(self.eri), ipovlp=p(self.ipovlp), dhcore=p(self.dhcore), eri_ip1=p(self.eri_ip1),
                        gnuc=p(self.gnuc), aoslices=p(self.aoslices))


@dataclass
class DeviceAOBatch:
    """AO integrals of ``count`` geometries of one molecule, stacked along a leading batch axis
    (include/evcont_hip.h ``evc_geometry_batch``)."""
    S: torch.Tensor          # (G,N,N)
    hcore: torch.Tensor      # (G,N,N)
    eri: torch.Tensor        # (G,N,N,N,N)
    enuc: torch.Tensor       # (G,)
    natm: int
    ipovlp: Optional[torch.Tensor] = None    # (G,3,N,N)
    dhcore: Optional[torch.Tensor] = None    # (G,A,3,N,N)
    eri_ip1: Optional[torch.Tensor] = None   # (G,3,N,N,N,N)
    gnuc: Optional[torch.Tensor] = None      # (G,A,3)
    aoslices: Optional[torch.Tensor] = None  # (A,2) int64, shared
    ip1_s2kl: bool = False                   # eri_ip1 is (G,3,N,N,N(N+1)/2) (EVC_FLAG_IP1_S2KL)
    eri_s4: bool = False                     # eri is (G,N(N+1)/2,N(N+1)/2) (EVC_FLAG_ERI_S4)

    @property
    def count(self) -> int:
        return int(self.S.shape[0])

    @property
    def nao(self) -> int:
        return int(self.S.shape[1])

    @staticmethod
    def stack(aos) -> "DeviceAOBatch":
        """Stack single-geometry ``DeviceAO`` objects (device-to-device copies)."""
        aos = list(aos)
        assert len({(bool(a.ip1_s2kl), bool(a.eri_s4)) for a in aos}) == 1, "mixed full / packed integrals in one batch"
        d = aos[0].S.device
        st = lambda name: (torch.stack([getattr(a, name) for a in aos]).contiguous()
                           if getattr(aos[0], name) is not None else None)
        return DeviceAOBatch(S=st("S"), hcore=st("hcore"), eri=st("eri"),
                             enuc=torch.tensor([a.enuc for a in aos], dtype=F64, device=d), natm=aos[0].natm,
                             ipovlp=st("ipovlp"), dhcore=st("dhcore"), eri_ip1=st("eri_ip1"), gnuc=st("gnuc"),
                             aoslices=aos[0].aoslices, ip1_s2kl=bool(aos[0].ip1_s2kl), eri_s4=bool(aos[0].eri_s4))

    @staticmethod
    def from_arrays(ao_list, device=None, energy_only: bool = False, pack_ip1: bool = False,
                    pack_eri: bool = False) -> "DeviceAOBatch":
        return DeviceAOBatch.stack([DeviceAO.from_arrays(a, device, energy_only, pack_ip1, pack_eri) for a in ao_list])

    def cstruct(self) -> "_lib.GeometryBatch":
        p = lambda t: (t.data_ptr() if t is not None else None)
        return _lib.GeometryBatch(natm=self.natm, count=self.count, enuc=p(self.enuc), S=p(self.S),
                                  hcore=p(self.hcore), eri=p(self.eri), ipovlp=p(self.ipovlp), dhcore=p(self.dhcore),
                                  eri_ip1=p(self.eri_ip1), gnuc=p(self.gnuc), aoslices=p(self.aoslices))


def _check_sym_first_call(ev, ao, energy_only: bool = False) -> None:
    """First call of an evaluator on the compressed layout (every call with EVCONT_AMD_CHECK_SYM=2): verify on the
    device, on the evaluator's stream, that the integrals have the symmetries the layout relies on."""
    if ev.t.layout != _lib.LAYOUT_SYM8 or _CHECK_SYM == "0" or (_CHECK_SYM != "2" and getattr(ev, "_sym_checked", False)):
        return
    if torch.cuda.is_current_stream_capturing():
        return
    st = ev.stream if ev.stream is not None else torch.cuda.current_stream(ev.t.device)
    with torch.cuda.stream(st):
        check_integral_symmetry(ao.eri, None if energy_only else ao.eri_ip1, ev.t.n, what=type(ev).__name__)
    ev._sym_checked = True


def _ip1_flag(trdms: "DeviceTRDMs", ao) -> int:
    """EVC_FLAG_IP1_S2KL / EVC_FLAG_ERI_S4 for geometries whose integrals are handed over packed."""
    f = (_lib.FLAG_IP1_S2KL if getattr(ao, "ip1_s2kl", False) else 0) | \
        (_lib.FLAG_ERI_S4 if getattr(ao, "eri_s4", False) else 0)
    if f and (trdms.layout != _lib.LAYOUT_SYM8 or trdms.n > 64):
        raise _lib.EvcontHipError("packed (s2kl / s4) integrals need training data in the compressed sym8 layout and "
                                  "N <= 64 (DeviceAO.from_arrays(..., pack_ip1=False, pack_eri=False) otherwise)")
    if trdms.n > 32 and f not in (0, _lib.FLAG_IP1_S2KL | _lib.FLAG_ERI_S4) and getattr(ao, "eri_ip1", None) is not None:
        raise _lib.EvcontHipError("N > 32: int2e and int2e_ip1 are handed over both packed (s4 and s2kl) or both full")
    return f


class BatchedEvaluator:
    """``count`` independent geometries per call (``evc_energy_with_grad_batch``): every launch covers
    the whole batch and the t-RDM is streamed once per up to 32 geometries.  Results stay on the device in
    ``energy (G,T)``, ``coeffs (G,T,T)``, ``grad (G,A,3)``."""

    def __init__(self, trdms: DeviceTRDMs, natm: int, count: int, stream: Optional["torch.cuda.Stream"] = None,
                 keep_density_matrices: bool = False, keep_hmat: bool = False, warm_start: bool = False,
                 keep_one_rdm: bool = False, energy_grad: Optional[torch.Tensor] = None):
        """``warm_start``: consecutive calls hold, slot by slot, nearby geometries (steps of ``count`` trajectories):
        from the second call on the eigensolvers start from the previous call's eigenvectors (EVC_FLAG_WARM_START).
        ``energy_grad``: caller's buffer of ``count * T + count * max(natm, 1) * 3`` doubles for the energies and the
        gradients -- e.g. PINNED HOST memory, which the device writes directly (no download copy, hosted.py)."""
        self.t, self.natm, self.count, self.stream = trdms, int(natm), int(count), stream
        self.warm_start, self._primed = bool(warm_start), False
        self.lib = _lib.load()
        d, n, T = trdms.device, trdms.n, trdms.T
        nbytes = self.lib.evc_workspace_bytes_batch(C.byref(trdms.cstruct), self.natm, self.count)
        if nbytes == 0:
            raise _lib.EvcontHipError("evc_workspace_bytes_batch: " + self.lib.evc_last_error().decode())
        # zero-filled once: the cached factorisation of S_train in it is recognised by content (a recycled allocator
        # block must not look like a hit)
        self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=d)
        self.ws_bytes = nbytes
        # (the library may attach a side stream to the workspace, include/evcont_hip.h evc_release_workspace)
        weakref.finalize(self, self.lib.evc_release_workspace, self.ws.data_ptr()).atexit = False
        G = self.count
        # energies and gradients share one buffer: a caller that wants both on the host fetches them with ONE copy
        na = max(self.natm, 1)
        if energy_grad is None:
            energy_grad = torch.zeros(G * T + G * na * 3, dtype=F64, device=d)
        assert energy_grad.dtype == F64 and energy_grad.numel() == G * T + G * na * 3 and energy_grad.is_contiguous()
        self.energy_grad = energy_grad
        self.energy = self.energy_grad[: G * T].view(G, T)
        self.coeffs = torch.zeros((G, T, T), dtype=F64, device=d)
        self.grad = self.energy_grad[G * T:].view(G, na, 3)
        self.d_pred = torch.zeros((G, n, n), dtype=F64, device=d) if (keep_density_matrices or keep_one_rdm) else None
        self.g_pred = torch.zeros((G, n, n, n, n), dtype=F64, device=d) if keep_density_matrices else None
        # the subspace Hamiltonians H(R) (lower triangles as handed to the eigensolver), for subset re-solves
        self.hmat = torch.zeros((G, T, T), dtype=F64, device=d) if keep_hmat else None
        p = lambda t: (t.data_ptr() if t is not None else None)
        self.out = _lib.OutputsBatch(energy=p(self.energy), coeffs=p(self.coeffs), grad=p(self.grad),
                                     d_pred=p(self.d_pred), g_pred=p(self.g_pred), hmat=p(self.hmat))

    def _sp(self) -> int:
        return self.stream.cuda_stream if self.stream is not None else _stream_ptr(self.t.device)

    def synchronize(self) -> None:
        (self.stream if self.stream is not None else torch.cuda.current_stream(self.t.device)).synchronize()

    def phase_loewdin(self, aob: DeviceAOBatch, stream: Optional["torch.cuda.Stream"] = None) -> None:
        """Loewdin orthogonalisation of the batch alone (reads S and hcore only): the next ``enqueue`` of the SAME
        geometries skips it (``EVC_FLAG_LOEWDIN_DONE``).  ``stream``: enqueue it there instead of on this
        evaluator's stream (the caller orders the streams)."""
        assert aob.count == self.count
        g = aob.cstruct()
        flags = _lib.FLAG_WARM_START if (self.warm_start and self._primed) else 0
        sp = stream.cuda_stream if stream is not None else self._sp()
        rc = self.lib.evc_phase_loewdin_batch(C.byref(self.t.cstruct), C.byref(g), flags, self.ws.data_ptr(),
                                              self.ws_bytes, sp)
        check(rc, "evc_phase_loewdin_batch")
        self._loewdin_done = True

    def enqueue(self, aob: DeviceAOBatch, nroots: int = 1, energy_only: bool = False) -> None:
        assert aob.count == self.count, "batch size is fixed at construction"
        _check_sym_first_call(self, aob, energy_only)
        g = aob.cstruct()
        flags = (_lib.FLAG_ENERGY_ONLY if energy_only else 0) | _ip1_flag(self.t, aob)
        if getattr(self, "_loewdin_done", False):
            flags |= _lib.FLAG_LOEWDIN_DONE
            self._loewdin_done = False
        if self.warm_start and self._primed:
            flags |= _lib.FLAG_WARM_START
        rc = self.lib.evc_energy_with_grad_batch(C.byref(self.t.cstruct), C.byref(g), C.byref(self.out), int(nroots),
                                                 flags, self.ws.data_ptr(), self.ws_bytes, self._sp())
        check(rc, "evc_energy_with_grad_batch")
        self._primed = True

    def energies_with_grads(self, aob: DeviceAOBatch):
        """(E[G], grad[G,A,3]) as numpy arrays."""
        self.enqueue(aob)
        self.synchronize()
        e = self.energy[:, 0].cpu().numpy().copy()
        if not np.all(np.isfinite(e)):
            raise np.linalg.LinAlgError("generalised eigenproblem failed for at least one geometry of the batch")
        return e, self.grad[:, : self.natm].cpu().numpy().copy()

    # -- phase API for the pair-sharded multi-GPU host (evcont_amd/distributed.py) -----------------
    def phase_hamiltonian(self, aob: DeviceAOBatch, rows_out: torch.Tensor) -> None:
        """Scaled two-body rows of this rank's pairs -> ``rows_out[g, :rows_local]`` (row stride = rows_out.stride(0))."""
        assert rows_out.dtype == F64 and rows_out.dim() == 2 and rows_out.shape[0] == self.count
        assert rows_out.stride(1) == 1 and rows_out.shape[1] >= self.t.rows_local
        _check_sym_first_call(self, aob)
        g = aob.cstruct()
        flags = _ip1_flag(self.t, aob) & _lib.FLAG_ERI_S4
        if getattr(self, "_loewdin_done", False):
            flags |= _lib.FLAG_LOEWDIN_DONE
            self._loewdin_done = False
        if self.warm_start and self._primed:
            flags |= _lib.FLAG_WARM_START
        rc = self.lib.evc_phase_hamiltonian_batch(C.byref(self.t.cstruct), C.byref(g), flags, rows_out.data_ptr(),
                                                  int(rows_out.stride(0)), self.ws.data_ptr(), self.ws_bytes, self._sp())
        check(rc, "evc_phase_hamiltonian_batch")

    def phase_solve(self, aob: DeviceAOBatch, rows_all: torch.Tensor, nroots: int = 1) -> None:
        assert rows_all.dtype == F64 and rows_all.dim() == 2 and rows_all.shape[0] == self.count
        assert rows_all.stride(1) == 1 and rows_all.shape[1] >= self.t.rows_total
        g = aob.cstruct()
        flags = _lib.FLAG_WARM_START if (self.warm_start and self._primed) else 0
        rc = self.lib.evc_phase_solve_batch(C.byref(self.t.cstruct), C.byref(g), rows_all.data_ptr(),
                                            int(rows_all.stride(0)), C.byref(self.out), int(nroots), flags,
                                            self.ws.data_ptr(), self.ws_bytes, self._sp())
        check(rc, "evc_phase_solve_batch")
        self._primed = True

    def phase_gradient(self, aob: DeviceAOBatch, partial_rank: bool) -> None:
        g = aob.cstruct()
        rc = self.lib.evc_phase_gradient_batch(C.byref(self.t.cstruct), C.byref(g), C.byref(self.out),
                                               (_lib.FLAG_PARTIAL_RANK if partial_rank else 0) | _ip1_flag(self.t, aob),
                                               self.ws.data_ptr(), self.ws_bytes, self._sp())
        check(rc, "evc_phase_gradient_batch")


class PipelinedBatchedEvaluator:
    """Batches submitted one after another by a caller that uses ONE stream, kept ``depth`` deep in flight inside the
    library: every batch runs on one of ``depth`` internal streams (each with its own workspace and result buffers),
    forked from the caller's stream at submission and joined to it only when its results are asked for.  The
    latency-bound single-workgroup kernels of one batch (Loewdin orthogonalisation, subspace solve, gradient tail:
    a quarter of a batch's time on 32 of 256 CUs) then run beside the chip-filling kernels of its neighbours -- what a
    caller would otherwise have to arrange with several streams of its own.

        pe = PipelinedBatchedEvaluator(trdms, natm, G)            # depth 3
        tickets = [pe.enqueue(aob) for aob in batches[:3]]
        for k, aob in enumerate(batches[3:]):
            r = pe.results(tickets[k])        # caller's stream now waits for THAT batch (no host synchronisation)
            ... consume r.energy / r.grad on the caller's stream ...
            tickets.append(pe.enqueue(aob))   # reuses the slot just consumed

    Ordering guarantees: a batch starts after everything the caller's stream held at ``enqueue`` (inputs uploaded
    asynchronously on that stream are complete), and after the previous batch of ITS slot has finished; a slot's
    buffers are overwritten ``depth`` submissions later, so results must be consumed (or ``results`` called) before
    then.  ``next_aob`` is accepted for compatibility with the round-2 interface and ignored."""

    def __init__(self, trdms: DeviceTRDMs, natm: int, count: int, stream: Optional["torch.cuda.Stream"] = None,
                 depth: int = 3, **kw):
        d = trdms.device
        self.device = d
        self._caller = stream          # None: torch's current stream at call time
        self.depth = max(1, int(depth))
        self.streams = [torch.cuda.Stream(d) for _ in range(self.depth)]
        self.evs = [BatchedEvaluator(trdms, natm, count, stream=self.streams[k], **kw) for k in range(self.depth)]
        self._submitted = [torch.cuda.Event() for _ in range(self.depth)]   # caller's stream at submission
        self._done = [torch.cuda.Event() for _ in range(self.depth)]        # internal stream: the slot's batch finished
        self._busy = [False] * self.depth
        self._k = 0

    def _cs(self) -> "torch.cuda.Stream":
        return self._caller if self._caller is not None else torch.cuda.current_stream(self.device)

    def enqueue(self, aob: DeviceAOBatch, next_aob: Optional[DeviceAOBatch] = None, nroots: int = 1,
                energy_only: bool = False) -> int:
        """Enqueue one batch; returns its slot (the ticket for ``results``)."""
        slot = self._k % self.depth
        self._k += 1
        st = self.streams[slot]
        self._submitted[slot].record(self._cs())
        st.wait_event(self._submitted[slot])                   # fork (the slot's previous batch precedes on `st` itself)
        self.evs[slot].enqueue(aob, nroots, energy_only)
        self._done[slot].record(st)
        self._busy[slot] = True
        return slot

    def results(self, slot: int) -> "BatchedEvaluator":
        """The evaluator holding the results of the batch submitted under this ticket; the caller's stream is made to
        wait for it (join)."""
        if self._busy[slot]:
            self._cs().wait_event(self._done[slot])
            self._busy[slot] = False
        return self.evs[slot]

    def synchronize(self) -> None:
        for st in self.streams:
            st.synchronize()
        self._cs().synchronize()


class ContinuationEvaluator:
    """Energy / energy+force of the continuation at one geometry per call
    (``get_energy_with_grad``, ``ab_initio_gradients_loewdin.py:308-379``)."""

    def __init__(self, trdms: DeviceTRDMs, natm: int, stream: Optional["torch.cuda.Stream"] = None,
                 warm_start: bool = False, want_two_rdm: bool = True):
        """``stream``: HIP stream every call of this evaluator is enqueued on (default: torch's current
        stream at call time).  Several evaluators on different streams may share one ``DeviceTRDMs``:
        each owns its workspace and outputs, so independent geometries overlap on the device.

        ``warm_start``: consecutive calls are steps of one trajectory (``EVC_FLAG_WARM_START``): from the
        second call on, the two Jacobi eigensolvers start from the eigenvectors the previous call left in the
        workspace.  Same results to solver tolerance (~1e-14), not bit for bit."""
        self.t = trdms
        self.natm = int(natm)
        self.stream = stream
        self.warm_start = bool(warm_start)
        self._primed = False
        self.lib = _lib.load()
        d, n, T = trdms.device, trdms.n, trdms.T
        nbytes = self.lib.evc_workspace_bytes(C.byref(trdms.cstruct), self.natm)
        if nbytes == 0:
            raise _lib.EvcontHipError("evc_workspace_bytes: " + self.lib.evc_last_error().decode())
        self.ws = torch.zeros(nbytes, dtype=torch.uint8, device=d)   # (zero-filled: see BatchedEvaluator)
        self.ws_bytes = nbytes
        weakref.finalize(self, self.lib.evc_release_workspace, self.ws.data_ptr()).atexit = False
        self.energy = torch.zeros(T, dtype=F64, device=d)
        self.coeffs = torch.zeros((T, T), dtype=F64, device=d)
        self.grad = torch.zeros((max(self.natm, 1), 3), dtype=F64, device=d)
        self.d_pred = torch.zeros((n, n), dtype=F64, device=d)
        self.g_pred = torch.zeros((n, n, n, n), dtype=F64, device=d) if want_two_rdm else None
        self.hmat = torch.zeros((T, T), dtype=F64, device=d)
        self.out = Outputs(energy=self.energy.data_ptr(), coeffs=self.coeffs.data_ptr(), grad=self.grad.data_ptr(),
                           d_pred=self.d_pred.data_ptr(),
                           g_pred=(self.g_pred.data_ptr() if want_two_rdm else None), hmat=self.hmat.data_ptr())

    def _sp(self) -> int:
        return self.stream.cuda_stream if self.stream is not None else _stream_ptr(self.t.device)

    def synchronize(self) -> None:
        (self.stream if self.stream is not None else torch.cuda.current_stream(self.t.device)).synchronize()

    # -- single-device fused path -------------------------------------------------------------
    def enqueue(self, ao: DeviceAO, nroots: int = 1, energy_only: bool = False) -> None:
        """Enqueue one evaluation on torch's current stream; no synchronisation."""
        _check_sym_first_call(self, ao, energy_only)
        g = ao.cstruct()
        flags = (_lib.FLAG_ENERGY_ONLY if energy_only else 0) | _ip1_flag(self.t, ao)
        if self.warm_start and self._primed:
            flags |= _lib.FLAG_WARM_START
        rc = self.lib.evc_energy_with_grad(C.byref(self.t.cstruct), C.byref(g), C.byref(self.out), int(nroots), flags,
                                           self.ws.data_ptr(), self.ws_bytes, self._sp())
        check(rc, "evc_energy_with_grad")
        self._primed = True

    def energy_with_grad(self, ao: DeviceAO, return_density_matrices: bool = False):
        self.enqueue(ao, 1, False)
        self.synchronize()
        e = float(self.energy[0].item())
        self._raise_if_nan(e)
        g = self.grad[: self.natm].cpu().numpy().copy()
        if return_density_matrices:
            if self.g_pred is None:
                raise _lib.EvcontHipError("this evaluator was built with want_two_rdm=False")
            return e, g, self.d_pred.cpu().numpy().copy(), self.g_pred.cpu().numpy().copy()
        return e, g

    def energies(self, ao: DeviceAO, nroots: int = 1):
        """Lowest ``nroots`` total energies and their coefficient vectors (rows)."""
        self.enqueue(ao, nroots, True)
        self.synchronize()
        e = self.energy[:nroots].cpu().numpy().copy()
        self._raise_if_nan(e[0])
        return e, self.coeffs.reshape(-1)[: nroots * self.t.T].reshape(nroots, self.t.T).cpu().numpy().copy()

    @staticmethod
    def _raise_if_nan(e):
        if not np.isfinite(e):
            # scipy.linalg.eigh raises LinAlgError when S is not positive definite (evcont.py:75)
            raise np.linalg.LinAlgError("generalised eigenproblem failed: overlap matrix not positive definite "
                                        "or non-finite input")

    # -- phase API for the pair-sharded multi-GPU host (evcont_amd/distributed.py) -----------------
    def phase_hamiltonian(self, ao: DeviceAO) -> torch.Tensor:
        """Returns a view of this rank's scaled two-body rows (length rows_local) in the workspace."""
        _check_sym_first_call(self, ao)
        g = ao.cstruct()
        p_rows, p_h1 = C.c_void_p(), C.c_void_p()
        flags = _ip1_flag(self.t, ao) & _lib.FLAG_ERI_S4
        if self.warm_start and self._primed:
            flags |= _lib.FLAG_WARM_START
        rc = self.lib.evc_phase_hamiltonian(C.byref(self.t.cstruct), C.byref(g), flags, self.ws.data_ptr(),
                                            self.ws_bytes, C.byref(p_rows), C.byref(p_h1), self._sp())
        check(rc, "evc_phase_hamiltonian")
        off = p_rows.value - self.ws.data_ptr()
        return self.ws[off: off + 8 * self.t.rows_local].view(F64)

    def phase_solve(self, ao: DeviceAO, rows_all: torch.Tensor, nroots: int = 1) -> None:
        g = ao.cstruct()
        assert rows_all.dtype == F64 and rows_all.numel() == self.t.rows_total and rows_all.is_contiguous()
        flags = _lib.FLAG_WARM_START if (self.warm_start and self._primed) else 0
        rc = self.lib.evc_phase_solve(C.byref(self.t.cstruct), C.byref(g), rows_all.data_ptr(), C.byref(self.out),
                                      int(nroots), flags, self.ws.data_ptr(), self.ws_bytes, self._sp())
        check(rc, "evc_phase_solve")
        self._primed = True

    def phase_set_coeffs(self, coeffs: torch.Tensor) -> None:
        """Row weights of the predicted RDMs from a caller-supplied coefficient vector (``evc_phase_set_coeffs``)."""
        assert coeffs.dtype == F64 and coeffs.numel() == self.t.T and coeffs.is_contiguous()
        rc = self.lib.evc_phase_set_coeffs(C.byref(self.t.cstruct), coeffs.data_ptr(), self.natm, self.ws.data_ptr(),
                                           self.ws_bytes, self._sp())
        check(rc, "evc_phase_set_coeffs")

    def energy_with_grad_nonhermitian(self, ao: DeviceAO, return_density_matrices: bool = False):
        """``get_energy_with_grad(..., hermitian=False)`` (``ab_initio_gradients_loewdin.py:341-379`` with the ``eig``
        branch of ``ab_initio_eigenvector_continuation.py:76-88``): the device assembles H(R) as for the Hermitian
        branch; the T x T non-symmetric pencil goes to ``scipy.linalg.eig`` on the host, as in the reference (pair
        layouts: upper triangle filled from the lower one, ``|Im| < 1e-5`` filter, ``argmin``), and ITS 2-norm
        eigenvector -- real part, as the reference takes it -- defines the predicted RDMs and the gradient."""
        import scipy.linalg
        if self.t.layout == _lib.LAYOUT_SYM8:
            raise _lib.EvcontHipError("hermitian=False needs the training data in the layout the caller holds, not sym8")
        if (self.t.row_offset, self.t.rows_local) != (0, self.t.rows_total):
            raise _lib.EvcontHipError("hermitian=False needs the complete t-RDM on this device")
        rows = self.phase_hamiltonian(ao)
        self.phase_solve(ao, rows, 1)
        self.synchronize()
        H = self.hmat.cpu().numpy().copy()
        if self.t.layout in (5, 2):
            iu = np.triu_indices(self.t.T)
            H[iu] = H.T[iu]
        vals, vecs = scipy.linalg.eig(H, self.t.S.cpu().numpy())
        valid = np.abs(vals.imag) < 1.0e-5
        k = int(np.argmin(vals[valid].real))
        e = float(vals[valid][k].real)
        vec = np.ascontiguousarray(vecs[:, valid][:, k].real, dtype=np.float64)
        self.phase_set_coeffs(torch.from_numpy(vec).to(self.t.device))
        self.phase_gradient(ao, False)
        self.synchronize()
        self._primed = False      # the workspace no longer holds a converged Hermitian solve
        g = self.grad[: self.natm].cpu().numpy().copy()
        if return_density_matrices:
            if self.g_pred is None:
                raise _lib.EvcontHipError("this evaluator was built with want_two_rdm=False")
            return e + ao.enuc, g, self.d_pred.cpu().numpy().copy(), self.g_pred.cpu().numpy().copy()
        return e + ao.enuc, g

    def phase_gradient(self, ao: DeviceAO, partial_rank: bool) -> None:
        g = ao.cstruct()
        rc = self.lib.evc_phase_gradient(C.byref(self.t.cstruct), C.byref(g), C.byref(self.out),
                                         (_lib.FLAG_PARTIAL_RANK if partial_rank else 0) | _ip1_flag(self.t, ao),
                                         self.ws.data_ptr(), self.ws_bytes, self._sp())
        check(rc, "evc_phase_gradient")
