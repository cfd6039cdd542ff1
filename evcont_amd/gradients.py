"""Device helper behind ``get_grad_elec_OAO`` (``ab_initio_gradients_loewdin.py:255-305``):
electronic gradient of GIVEN one-/two-body RDMs in the Loewdin-orthogonalised AO basis."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import check
from .evaluator import DeviceAO, F64, _stream_ptr


def grad_elec_oao_device(ao: DeviceAO, one_rdm: torch.Tensor, two_rdm: torch.Tensor) -> torch.Tensor:
    """(A,3) electronic gradient; ``one_rdm`` (N,N) and ``two_rdm`` (N,N,N,N) are device tensors."""
    lib = _lib.load()
    n, natm = ao.nao, ao.natm
    d = ao.S.device
    one_rdm = one_rdm.to(d, F64).contiguous()
    two_rdm = two_rdm.to(d, F64).contiguous()
    assert tuple(one_rdm.shape) == (n, n) and tuple(two_rdm.shape) == (n, n, n, n)
    nbytes = lib.evc_grad_elec_ws_bytes(n, natm)
    if nbytes == 0:
        raise _lib.EvcontHipError(f"evc_grad_elec_ws_bytes: unsupported size n={n}")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
    grad = torch.empty((natm, 3), dtype=F64, device=d)
    g = ao.cstruct()
    check(lib.evc_grad_elec_oao(n, C.byref(g), one_rdm.data_ptr(), two_rdm.data_ptr(), grad.data_ptr(),
                                ws.data_ptr(), nbytes, _stream_ptr(d)), "evc_grad_elec_oao")
    return grad
