"""Training-data containers: the array semantics shared by the reference's three container classes
(``FCI_EVCont.py:106-151``, ``CASCI_EVCont.py:183-202,337-361``, ``DMRG_EVCont.py:60-87,451-496``).

``overlap (T,T)``, ``one_rdm (T,T,N,N)`` and ``two_rdm (T,T,N,N,N,N)`` are C-order float64 arrays that
grow by one row + one column per training state; the new bra/ket blocks ``[-1, i]`` and ``[i, -1]``
both receive the SAME matrix ``<new| . |i>`` (the reference stores ``.conj()``, not the transpose),
and ``prune_datapoints(keep_ids)`` is ``np.ix_`` fancy indexing on the two leading axes.

On top of that the containers keep a device-resident copy in step (``device_trdms``): the evaluator
streams the two-body t-RDM from HBM twice per geometry, so it is uploaded once per change of the
training set — in the electron-pair-packed ``(P, M)`` layout by default, a quarter of the bytes of the
six-index array (``ab_initio_eigenvector_continuation.py:59-68``) — and reused by every evaluation.
"""
from __future__ import annotations

from typing import Optional, Sequence

import numpy as np


def grow_trdms(overlap, one_rdm, two_rdm, ovlp_row, one_rows, two_rows):
    """Append one training state.  ``ovlp_row[i]``, ``one_rows[i]``, ``two_rows[i]`` are the overlap and the
    transition RDMs between the NEW state (bra) and state ``i`` (ket), ``i = 0..T`` with ``i = T`` the new
    state itself.  Returns the grown ``(overlap, one_rdm, two_rdm)`` (new arrays; inputs untouched)."""
    ovlp_row = np.asarray(ovlp_row, dtype=np.float64)
    one_rows = np.asarray(one_rows, dtype=np.float64)
    two_rows = np.asarray(two_rows, dtype=np.float64)
    T = ovlp_row.shape[0] - 1
    n = one_rows.shape[-1]
    assert one_rows.shape == (T + 1, n, n) and two_rows.shape == (T + 1, n, n, n, n)
    assert (overlap is None and T == 0) or (overlap is not None and overlap.shape[0] == T)
    ov = np.ones((T + 1, T + 1))
    o1 = np.ones((T + 1, T + 1, n, n))
    o2 = np.ones((T + 1, T + 1, n, n, n, n))
    if overlap is not None:
        ov[:-1, :-1] = overlap
        o1[:-1, :-1] = one_rdm
        o2[:-1, :-1] = two_rdm
    ov[-1, :] = ovlp_row
    ov[:, -1] = ovlp_row
    o1[-1, :] = one_rows
    o1[:, -1] = one_rows
    o2[-1, :] = two_rows
    o2[:, -1] = two_rows
    return ov, o1, o2


class TRDMContainer:
    """Base of the three container classes: the arrays, pruning and the device-resident copy."""

    def __init__(self):
        self.overlap: Optional[np.ndarray] = None
        self.one_rdm: Optional[np.ndarray] = None
        self.two_rdm: Optional[np.ndarray] = None
        self._device = None
        self._device_key = None

    # -- growth / pruning ----------------------------------------------------------------------
    def _append_state(self, ovlp_row, one_rows, two_rows) -> None:
        self.overlap, self.one_rdm, self.two_rdm = grow_trdms(self.overlap, self.one_rdm, self.two_rdm,
                                                              ovlp_row, one_rows, two_rows)
        self._invalidate()

    def _prune_arrays(self, keep_ids: Sequence[int]) -> None:
        keep_ids = list(keep_ids)
        if self.overlap is not None:
            self.overlap = self.overlap[np.ix_(keep_ids, keep_ids)]
        if self.one_rdm is not None:
            self.one_rdm = self.one_rdm[np.ix_(keep_ids, keep_ids)]
        if self.two_rdm is not None:
            self.two_rdm = self.two_rdm[np.ix_(keep_ids, keep_ids)]
        self._invalidate()

    def _invalidate(self) -> None:
        """The training set changed: drop the device copies made from the old arrays (this container's and the
        upload cache of the mol-level API, which is keyed on a fingerprint of the arrays it was given)."""
        from . import cache
        self._device, self._device_key = None, None
        cache.clear()

    def prune_datapoints(self, keep_ids) -> None:
        self._prune_arrays(keep_ids)

    @property
    def ntrain(self) -> int:
        return 0 if self.overlap is None else int(self.overlap.shape[0])

    # -- device-resident copy ------------------------------------------------------------------
    def device_trdms(self, layout: str = "pack2", device=None):
        """``evaluator.DeviceTRDMs`` of the current training set, uploaded on first use and whenever the
        arrays were replaced (append, prune, or a script assigning ``np.load`` results to the attributes,
        ``md_H30_evcont_from_DMRG.py:72-85``).  ``layout``: "pack2" (pairs x packed electrons, needs the
        bra<->ket symmetric data every container of the reference produces), "pair5", "elec3", "full6", or "sym8"
        (8-fold compressed on the device from the pack2 form, ``DeviceTRDMs.compress_sym8_``)."""
        from .evaluator import DeviceTRDMs
        from .synthetic import pack_rows
        if self.two_rdm is None:
            raise ValueError("the container holds no training data yet")
        from .cache import key_of
        key = key_of(self.one_rdm, self.two_rdm, self.overlap, (layout, str(device)))   # address + content sample
        if self._device is None or self._device_key != key:
            pairs, elec = {"full6": (False, False), "pair5": (True, False), "elec3": (False, True),
                           "pack2": (True, True), "sym8": (True, True)}[layout]
            two = np.asarray(self.two_rdm, dtype=np.float64)
            if two.ndim == 6 and (pairs or elec):
                two = pack_rows(two, pairs, elec)
            self._device = DeviceTRDMs(self.one_rdm, two, self.overlap, device,
                                       compress="sym8" if layout == "sym8" else None)
            self._device_key = key
        return self._device
