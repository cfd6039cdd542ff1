"""Device helpers behind the tensor-valued building blocks of
``ab_initio_gradients_loewdin.py`` and ``get_grad_elec_OAO`` (:255-305).  Everything takes and
returns device tensors (float64, C order, the reference's index order)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import check
from .evaluator import DeviceAO, F64, _stream_ptr


def _c(t: torch.Tensor, d) -> torch.Tensor:
    return t.to(d, F64).contiguous()


def grad_elec_oao_device(ao: DeviceAO, one_rdm: torch.Tensor, two_rdm: torch.Tensor,
                         trafo: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(A,3) electronic gradient of given RDMs (N,N), (N,N,N,N); ``trafo`` = caller's ao_mo_trafo."""
    lib = _lib.load()
    n, natm = ao.nao, ao.natm
    d = ao.S.device
    one_rdm, two_rdm = _c(one_rdm, d), _c(two_rdm, d)
    assert tuple(one_rdm.shape) == (n, n) and tuple(two_rdm.shape) == (n, n, n, n)
    if trafo is not None:
        trafo = _c(trafo, d)
        assert tuple(trafo.shape) == (n, n)
    nbytes = lib.evc_grad_elec_ws_bytes(n, natm)
    if nbytes == 0:
        raise _lib.EvcontHipError(f"evc_grad_elec_ws_bytes: unsupported size n={n}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
    grad = torch.empty((natm, 3), dtype=F64, device=d)
    g = ao.cstruct()
    check(lib.evc_grad_elec_oao(n, C.byref(g), trafo.data_ptr() if trafo is not None else None,
                                one_rdm.data_ptr(), two_rdm.data_ptr(), grad.data_ptr(), ws.data_ptr(), nbytes,
                                _stream_ptr(d)), "evc_grad_elec_oao")
    return grad


def loewdin_trafo_grad_device(S: torch.Tensor) -> torch.Tensor:
    """LG[p,q,a,b] = d X_pq / d S_ab(sym)  (N,N,N,N)."""
    lib = _lib.load()
    n = S.shape[0]
    S = S.contiguous()
    LG = torch.empty((n, n, n, n), dtype=F64, device=S.device)
    ws = torch.empty(2 * n * n + n, dtype=F64, device=S.device)
    check(lib.evc_loewdin_trafo_grad(S.data_ptr(), n, LG.data_ptr(), ws.data_ptr(), _stream_ptr(S.device)),
          "evc_loewdin_trafo_grad")
    return LG


def derivative_ao_mo_trafo_device(ao: DeviceAO) -> torch.Tensor:
    """dX[k,l,A,x]  (N,N,A,3)."""
    lib = _lib.load()
    n, natm, d = ao.nao, ao.natm, ao.S.device
    dX = torch.empty((n, n, natm, 3), dtype=F64, device=d)
    ws = torch.empty(2 * n * n + n, dtype=F64, device=d)
    check(lib.evc_derivative_ao_mo_trafo(ao.S.data_ptr(), ao.ipovlp.data_ptr(), ao.aoslices.data_ptr(), n, natm,
                                         dX.data_ptr(), ws.data_ptr(), _stream_ptr(d)), "evc_derivative_ao_mo_trafo")
    return dX


def one_el_grad_device(ao: DeviceAO, X: torch.Tensor, dX: torch.Tensor) -> torch.Tensor:
    """h1_jac[j,n,A,x]  (N,N,A,3)."""
    lib = _lib.load()
    n, natm, d = ao.nao, ao.natm, ao.S.device
    X, dX = _c(X, d), _c(dX, d)
    out = torch.empty((n, n, natm, 3), dtype=F64, device=d)
    check(lib.evc_one_el_grad(X.data_ptr(), ao.hcore.data_ptr(), ao.dhcore.data_ptr(), dX.data_ptr(), n, natm,
                              out.data_ptr(), _stream_ptr(d)), "evc_one_el_grad")
    return out


def two_el_grad_device(h2_ao, two_rdm, X, dX, ip1, aoslices: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    d = h2_ao.device
    n, natm = X.shape[0], int(aoslices.shape[0])
    h2_ao, two_rdm, X, dX, ip1 = (_c(t, d) for t in (h2_ao, two_rdm, X, dX, ip1))
    nbytes = lib.evc_two_el_grad_ws_bytes(n)
    if nbytes == 0:
        raise _lib.EvcontHipError(f"evc_two_el_grad_ws_bytes: unsupported size n={n}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
    out = torch.empty((natm, 3), dtype=F64, device=d)
    check(lib.evc_two_el_grad(h2_ao.data_ptr(), two_rdm.data_ptr(), X.data_ptr(), dX.data_ptr(), ip1.data_ptr(),
                              aoslices.data_ptr(), n, natm, out.data_ptr(), ws.data_ptr(), nbytes, _stream_ptr(d)),
          "evc_two_el_grad")
    return out


def contract_nnA3_device(T: torch.Tensor, M: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """out[A,x] = sum_ij T[i,j,A,x] M[i,j] (or M[j,i])."""
    lib = _lib.load()
    d = T.device
    n, natm = T.shape[0], T.shape[2]
    T, M = _c(T, d), _c(M, d)
    out = torch.empty((natm, 3), dtype=F64, device=d)
    check(lib.evc_contract_nnA3(T.data_ptr(), M.data_ptr(), int(transposed), n, natm, out.data_ptr(), _stream_ptr(d)),
          "evc_contract_nnA3")
    return out
