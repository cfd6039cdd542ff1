"""On-disk training data -> device-resident packed t-RDMs.

Two formats occur in the reference's scripts:

* one directory per training pair ``MPS_cross_{a}_{b}/`` (``a >= b``) holding ``ovlp.npy`` (scalar),
  ``one_rdm.npy (N,N)`` and ``two_rdm.npy (M,)`` — the electron-pair-packed two-body t-RDM of that pair —
  written by ``03_Zundel_continuation_evaluate_MPS_t_RDMs.py:110`` and assembled by
  ``04_Zundel_continuation_MD.py:99-128`` into ``overlap (T,T)``, ``one_rdm (T,T,N,N)`` (upper blocks
  = the untransposed lower ones) and ``two_rdm (P,M)``;
* the checkpoints ``overlap.npy / one_rdm.npy / two_rdm.npy`` of the containers and of
  ``converge_EVCont_MD`` (``MD_utils.py:176-184``), the two-body array six-index ``(T,T,N,N,N,N)``.

Both are streamed row by row into the ``(P, ld)`` matrix the evaluator contracts (pairs in
``np.tril_indices`` order, packed electrons): the multi-GB two-body array never exists as a whole on
the host (``np.load(mmap_mode="r")``), and a six-index checkpoint is packed on the device pair by pair.
``prefix`` gives the training set restricted to its first ``k`` states — the pairs of the first ``k``
states are exactly the first ``k(k+1)/2`` rows, the slicing rule of
``05_Zundel_test_potential_energy.py:114-131`` — without copying the two-body rows.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import ops
from .evaluator import DeviceTRDMs, F64, _dev, _pad_even, layout_shape
from ._lib import TrdmSet

PAIR_DIR = "MPS_cross_{}_{}"


def save_pair_directories(root: str, overlap, one_rdm, two_rdm, pattern: str = PAIR_DIR) -> None:
    """Write the per-pair layout.  ``two_rdm``: ``(P,M)`` packed, or six-index (packed here with multiplier 1)."""
    overlap, one_rdm, two_rdm = np.asarray(overlap), np.asarray(one_rdm), np.asarray(two_rdm)
    T, n = overlap.shape[0], one_rdm.shape[-1]
    ia, ib = np.tril_indices(T)
    r, c = np.tril_indices(n * n)
    for p, (a, b) in enumerate(zip(ia, ib)):
        d = os.path.join(root, pattern.format(a, b))
        os.makedirs(d, exist_ok=True)
        np.save(os.path.join(d, "ovlp.npy"), overlap[a, b])
        np.save(os.path.join(d, "one_rdm.npy"), one_rdm[a, b])
        row = two_rdm[p] if two_rdm.ndim == 2 else two_rdm[a, b].reshape(n * n, n * n)[r, c]
        np.save(os.path.join(d, "two_rdm.npy"), np.ascontiguousarray(row))


def load_pair_directories(root: str, ntrain: int, device=None, pattern: str = PAIR_DIR) -> DeviceTRDMs:
    """Assemble the training set of ``04_Zundel_continuation_MD.py:99-128`` directly on the device."""
    dev = _dev(device)
    T = int(ntrain)
    ia, ib = np.tril_indices(T)
    first = np.load(os.path.join(root, pattern.format(0, 0), "one_rdm.npy"))
    n = int(first.shape[-1])
    rows, cols = layout_shape(2, T, n)
    ld = (cols + 15) // 16 * 16
    two = torch.zeros((rows, ld), dtype=F64, device=dev)
    S = np.zeros((T, T))
    one = np.zeros((T, T, n, n))
    for p, (a, b) in enumerate(zip(ia, ib)):
        d = os.path.join(root, pattern.format(a, b))
        S[a, b] = S[b, a] = float(np.load(os.path.join(d, "ovlp.npy")))
        o = np.load(os.path.join(d, "one_rdm.npy"))
        one[a, b] = one[b, a] = o                      # upper blocks = untransposed lower ones (:113-117)
        row = np.load(os.path.join(d, "two_rdm.npy"), mmap_mode="r")
        assert row.shape == (cols,), f"{d}/two_rdm.npy has shape {row.shape}, expected ({cols},)"
        two[p, :cols].copy_(torch.from_numpy(np.array(row, dtype=np.float64)))
    return _from_padded(one, S, two, T, n)


def _from_padded(one, S, two_padded: torch.Tensor, T: int, n: int) -> DeviceTRDMs:
    """DeviceTRDMs around an already padded (P, ld) device matrix (no further copy)."""
    self = DeviceTRDMs.__new__(DeviceTRDMs)
    dev = two_padded.device
    rows, cols = layout_shape(2, T, n)
    self.device, self.T, self.n, self.layout = dev, T, n, 2
    self.two = two_padded
    self.rows_total, self.cols, self.ld = rows, cols, int(two_padded.shape[1])
    self.row_offset, self.rows_local = 0, rows
    as_t = lambda x: (x if torch.is_tensor(x) else torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)))
    self.one = _pad_even(as_t(one).to(dev, F64).reshape(T * T, n * n))
    self.S = as_t(S).to(dev, F64).contiguous()
    self.cstruct = TrdmSet(n=n, ntrain=T, layout=2, reserved=0, rows2=rows, row_offset=0, rows2_total=rows,
                           cols2=cols, ld2=self.ld, ld1=self.one.shape[1], two_rdm=self.two.data_ptr(),
                           one_rdm=self.one.data_ptr(), s_train=self.S.data_ptr())
    return self


def load_checkpoint(directory: str, device=None, suffix: str = "") -> DeviceTRDMs:
    """``overlap{suffix}.npy / one_rdm{suffix}.npy / two_rdm{suffix}.npy`` -> packed device t-RDMs.  The
    two-body file is memory-mapped; a six-index array is packed on the device one pair block (N^4 doubles)
    at a time, a ``(P,M)`` array is copied row by row."""
    dev = _dev(device)
    S = np.load(os.path.join(directory, f"overlap{suffix}.npy"))
    one = np.load(os.path.join(directory, f"one_rdm{suffix}.npy"))
    two = np.load(os.path.join(directory, f"two_rdm{suffix}.npy"), mmap_mode="r")
    T, n = int(S.shape[0]), int(one.shape[-1])
    rows, cols = layout_shape(2, T, n)
    ld = (cols + 15) // 16 * 16
    out = torch.zeros((rows, ld), dtype=F64, device=dev)
    ia, ib = np.tril_indices(T)
    if two.ndim == 2:
        assert two.shape == (rows, cols)
        for p in range(rows):
            out[p, :cols].copy_(torch.from_numpy(np.array(two[p], dtype=np.float64)))
    else:
        assert two.shape == (T, T, n, n, n, n)
        for p, (a, b) in enumerate(zip(ia, ib)):
            block = torch.from_numpy(np.array(two[a, b], dtype=np.float64)).to(dev)
            out[p, :cols].copy_(ops.pack_pair_sym(block, 1.0))
    return _from_padded(one, S, out, T, n)


def prefix(t: DeviceTRDMs, ntrain: int) -> DeviceTRDMs:
    """The first ``ntrain`` training states of a pair-layout set (pack2, pair5 or the compressed sym8); shares the
    two-body rows with ``t``."""
    k = int(ntrain)
    assert t.layout in (2, 5, 8) and 1 <= k <= t.T and t.row_offset == 0 and t.rows_local == t.rows_total
    self = DeviceTRDMs.__new__(DeviceTRDMs)
    rows, cols = layout_shape(t.layout, k, t.n)
    self.device, self.T, self.n, self.layout = t.device, k, t.n, t.layout
    self.two = t.two[:rows]
    self.rows_total, self.cols, self.ld = rows, cols, t.ld
    self.row_offset, self.rows_local = 0, rows
    n2 = t.n * t.n
    one = t.one[:, :n2].reshape(t.T, t.T, n2)[:k, :k].reshape(k * k, n2)
    self.one = _pad_even(one.contiguous())
    self.S = t.S[:k, :k].contiguous()
    self.cstruct = TrdmSet(n=t.n, ntrain=k, layout=t.layout, reserved=0, rows2=rows, row_offset=0, rows2_total=rows,
                           cols2=cols, ld2=t.ld, ld1=self.one.shape[1], two_rdm=self.two.data_ptr(),
                           one_rdm=self.one.data_ptr(), s_train=self.S.data_ptr())
    self._parent = t          # keeps the shared rows alive
    return self
