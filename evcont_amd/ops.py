"""Thin tensor-level wrappers of the individual C-ABI kernels (device tensors in, device
tensors out, enqueued on torch's current stream).  Used by the API mirror modules and by
the parity tests; the per-geometry hot path goes through ``evaluator.ContinuationEvaluator``
(one C call per geometry) instead."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check

F64 = torch.float64


def _s(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _chk(t: torch.Tensor, name: str) -> torch.Tensor:
    if not (torch.is_tensor(t) and t.is_cuda and t.dtype == F64 and t.is_contiguous()):
        raise ValueError(f"{name}: expected a contiguous float64 device tensor")
    return t


def to_device(x, device) -> torch.Tensor:
    if torch.is_tensor(x):
        return x.to(device, F64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(device)


def padded_matrix(mat: torch.Tensor) -> torch.Tensor:
    """(rows, cols) -> zero-padded (rows, ld) with ld a multiple of 16 doubles."""
    rows, cols = mat.shape
    ld = (cols + 15) // 16 * 16
    out = torch.zeros((rows, ld), dtype=F64, device=mat.device)
    out[:, :cols].copy_(mat)
    return out


def gemv_rows(A: torch.Tensor, cols: int, v: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
    """y = alpha * A[:, :cols] @ v ; A is a (rows, ld) padded matrix."""
    lib = _lib.load()
    _chk(A, "A"), _chk(v, "v")
    rows, ld = A.shape
    ws_bytes = lib.evc_gemv_rows_ws_bytes(rows, cols)
    ws = torch.empty(max(ws_bytes, 8), dtype=torch.uint8, device=A.device)
    y = torch.empty(rows, dtype=F64, device=A.device)
    check(lib.evc_gemv_rows(A.data_ptr(), rows, cols, ld, v.data_ptr(), float(alpha), y.data_ptr(), ws.data_ptr(),
                            ws_bytes, _s(A)), "evc_gemv_rows")
    return y


def gemv_cols(A: torch.Tensor, cols: int, w: torch.Tensor) -> torch.Tensor:
    """out = w @ A[:, :cols]."""
    lib = _lib.load()
    _chk(A, "A"), _chk(w, "w")
    rows, ld = A.shape
    out = torch.empty(cols + (cols & 1), dtype=F64, device=A.device)
    check(lib.evc_gemv_cols(A.data_ptr(), rows, cols, ld, w.data_ptr(), out.data_ptr(), _s(A)), "evc_gemv_cols")
    return out[:cols]


def pack_pair_sym(h2: torch.Tensor, diag_multiplier: float = 1.0, pad_to: int = 0) -> torch.Tensor:
    lib = _lib.load()
    _chk(h2, "h2")
    n = h2.shape[0]
    M = n * n * (n * n + 1) // 2
    L = max(M, pad_to)
    out = torch.empty(L, dtype=F64, device=h2.device)
    check(lib.evc_pack_pair_sym(h2.data_ptr(), n, float(diag_multiplier), out.data_ptr(), L, _s(h2)),
          "evc_pack_pair_sym")
    return out


def unpack_pair_sym(v: torch.Tensor, norb: int) -> torch.Tensor:
    lib = _lib.load()
    _chk(v, "v")
    out = torch.empty((norb,) * 4, dtype=F64, device=v.device)
    check(lib.evc_unpack_pair_sym(v.data_ptr(), norb, out.data_ptr(), _s(v)), "evc_unpack_pair_sym")
    return out


def quarter_transform(t: torch.Tensor, Cm: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    lib = _lib.load()
    _chk(t, "t"), _chk(Cm, "C")
    n = Cm.shape[0]
    out = torch.empty_like(t)
    check(lib.evc_quarter_transform(t.data_ptr(), Cm.data_ptr(), int(transposed), n, out.data_ptr(), _s(t)),
          "evc_quarter_transform")
    return out


def four_index_transform(t: torch.Tensor, Cm: torch.Tensor, transposed: bool = False, want_three_quarter=False):
    lib = _lib.load()
    _chk(t, "t"), _chk(Cm, "C")
    n = Cm.shape[0]
    out, tmp = torch.empty_like(t), torch.empty_like(t)
    k3 = torch.empty_like(t) if want_three_quarter else None
    check(lib.evc_four_index_transform(t.data_ptr(), Cm.data_ptr(), int(transposed), n, out.data_ptr(),
                                       tmp.data_ptr(), k3.data_ptr() if k3 is not None else None, _s(t)),
          "evc_four_index_transform")
    return (out, k3) if want_three_quarter else out


def loewdin(S: torch.Tensor, hcore: torch.Tensor = None):
    """X, U, s[, h1] of S = U diag(s) U^T."""
    lib = _lib.load()
    _chk(S, "S")
    n = S.shape[0]
    X, U = torch.empty_like(S), torch.empty_like(S)
    s = torch.empty(n, dtype=F64, device=S.device)
    h1 = torch.empty_like(S) if hcore is not None else None
    check(lib.evc_loewdin(S.data_ptr(), _chk(hcore, "hcore").data_ptr() if hcore is not None else None, n,
                          X.data_ptr(), U.data_ptr(), s.data_ptr(), h1.data_ptr() if h1 is not None else None,
                          _s(S)), "evc_loewdin")
    return (X, U, s, h1) if hcore is not None else (X, U, s)


def subspace_solve(h1rows: torch.Tensor, h2rows: torch.Tensor, S: torch.Tensor, layout: int, nroots: int = 1,
                   e_shift: float = 0.0):
    """(evals[nroots], evecs[nroots,T], w2, w1, H)"""
    lib = _lib.load()
    _chk(h1rows, "h1rows"), _chk(h2rows, "h2rows"), _chk(S, "S")
    T = S.shape[0]
    d = S.device
    ev = torch.empty(nroots, dtype=F64, device=d)
    vec = torch.empty((nroots, T), dtype=F64, device=d)
    w2 = torch.empty(h2rows.numel(), dtype=F64, device=d)
    w1 = torch.empty(T * T, dtype=F64, device=d)
    H = torch.empty((T, T), dtype=F64, device=d)
    nws = lib.evc_subspace_solve_ws_bytes(T, 1)   # 0 for T <= 32
    ws = torch.empty(nws, dtype=torch.uint8, device=d) if nws else None
    check(lib.evc_subspace_solve(h1rows.data_ptr(), h2rows.data_ptr(), S.data_ptr(), T, layout, nroots,
                                 float(e_shift), ev.data_ptr(), vec.data_ptr(), w2.data_ptr(), w1.data_ptr(),
                                 H.data_ptr(), ws.data_ptr() if ws is not None else None, nws, _s(S)),
          "evc_subspace_solve")
    return ev, vec, w2, w1, H
