"""ctypes binding of ``libevcont_hip.so`` (C ABI declared in ``include/evcont_hip.h``).

There is no CPU fallback: if the library is missing or a call fails, an exception is
raised.  ``load()`` never builds implicitly on a machine without hipcc.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libevcont_hip.so")

LAYOUT_FULL6, LAYOUT_PAIR5, LAYOUT_ELEC3, LAYOUT_PACK2 = 6, 5, 3, 2
LAYOUT_SYM8 = 8   # device-side 8-fold compressed layout (include/evcont_hip.h EVC_LAYOUT_SYM8)
FLAG_ENERGY_ONLY, FLAG_PARTIAL_RANK, FLAG_WARM_START, FLAG_IP1_S2KL, FLAG_ERI_S4, FLAG_LOEWDIN_DONE = 1, 2, 4, 8, 16, 32
ABI_VERSION = 9

c_double_p = C.c_void_p  # device pointers travel as integers


class TrdmSet(C.Structure):
    _fields_ = [("n", C.c_int32), ("ntrain", C.c_int32), ("layout", C.c_int32), ("reserved", C.c_int32),
                ("rows2", C.c_int64), ("row_offset", C.c_int64), ("rows2_total", C.c_int64),
                ("cols2", C.c_int64), ("ld2", C.c_int64), ("ld1", C.c_int64),
                ("two_rdm", C.c_void_p), ("one_rdm", C.c_void_p), ("s_train", C.c_void_p)]


class Geometry(C.Structure):
    _fields_ = [("natm", C.c_int32), ("reserved", C.c_int32), ("enuc", C.c_double),
                ("S", C.c_void_p), ("hcore", C.c_void_p), ("eri", C.c_void_p), ("ipovlp", C.c_void_p),
                ("dhcore", C.c_void_p), ("eri_ip1", C.c_void_p), ("gnuc", C.c_void_p),
                ("aoslices", C.c_void_p)]


class Outputs(C.Structure):
    _fields_ = [("energy", C.c_void_p), ("coeffs", C.c_void_p), ("grad", C.c_void_p),
                ("d_pred", C.c_void_p), ("g_pred", C.c_void_p), ("hmat", C.c_void_p)]


class GeometryBatch(C.Structure):
    _fields_ = [("natm", C.c_int32), ("count", C.c_int32), ("enuc", C.c_void_p),
                ("S", C.c_void_p), ("hcore", C.c_void_p), ("eri", C.c_void_p), ("ipovlp", C.c_void_p),
                ("dhcore", C.c_void_p), ("eri_ip1", C.c_void_p), ("gnuc", C.c_void_p),
                ("aoslices", C.c_void_p)]


class OutputsBatch(C.Structure):
    _fields_ = [("energy", C.c_void_p), ("coeffs", C.c_void_p), ("grad", C.c_void_p),
                ("d_pred", C.c_void_p), ("g_pred", C.c_void_p), ("hmat", C.c_void_p)]


# symbol -> (restype, argtypes); also the list the CPU test checks against the header
SIGNATURES = {
    "evc_abi_version": (C.c_int, []),
    "evc_last_error": (C.c_char_p, []),
    "evc_gemv_rows_ws_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "evc_gemv_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_double,
                                C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_gemv_cols": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                C.c_void_p]),
    "evc_pack_pair_sym": (C.c_int, [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int64, C.c_void_p]),
    "evc_unpack_pair_sym": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "evc_quarter_transform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "evc_four_index_transform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "evc_loewdin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, C.c_void_p]),
    "evc_subspace_solve_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "evc_subspace_solve": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                     C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_subspace_solve_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_integrals_oao_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "evc_integrals_oao_batch": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_workspace_bytes": (C.c_size_t, [C.POINTER(TrdmSet), C.c_int]),
    "evc_phase_hamiltonian": (C.c_int, [C.POINTER(TrdmSet), C.POINTER(Geometry), C.c_int, C.c_void_p, C.c_size_t,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    "evc_phase_solve": (C.c_int, [C.POINTER(TrdmSet), C.POINTER(Geometry), C.c_void_p, C.POINTER(Outputs),
                                  C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_phase_gradient": (C.c_int, [C.POINTER(TrdmSet), C.POINTER(Geometry), C.POINTER(Outputs), C.c_int,
                                     C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_phase_set_coeffs": (C.c_int, [C.POINTER(TrdmSet), C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_energy_with_grad": (C.c_int, [C.POINTER(TrdmSet), C.POINTER(Geometry), C.POINTER(Outputs), C.c_int,
                                       C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_workspace_bytes_batch": (C.c_size_t, [C.POINTER(TrdmSet), C.c_int, C.c_int]),
    "evc_energy_with_grad_batch": (C.c_int, [C.POINTER(TrdmSet), C.POINTER(GeometryBatch), C.POINTER(OutputsBatch),
                                             C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_phase_loewdin_batch": (C.c_int, [C.POINTER(TrdmSet), C.POINTER(GeometryBatch), C.c_int, C.c_void_p, C.c_size_t,
                                          C.c_void_p]),
    "evc_phase_hamiltonian_batch": (C.c_int, [C.POINTER(TrdmSet), C.POINTER(GeometryBatch), C.c_int, C.c_void_p,
                                              C.c_int64, C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_phase_solve_batch": (C.c_int, [C.POINTER(TrdmSet), C.POINTER(GeometryBatch), C.c_void_p, C.c_int64,
                                        C.POINTER(OutputsBatch), C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "evc_phase_gradient_batch": (C.c_int, [C.POINTER(TrdmSet), C.POINTER(GeometryBatch), C.POINTER(OutputsBatch),
                                           C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_grad_elec_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "evc_grad_elec_oao": (C.c_int, [C.c_int, C.POINTER(Geometry), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_loewdin_trafo_grad": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "evc_derivative_ao_mo_trafo": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    "evc_one_el_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                  C.c_void_p]),
    "evc_two_el_grad_ws_bytes": (C.c_size_t, [C.c_int]),
    "evc_two_el_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "evc_contract_nnA3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "evc_profile_begin": (C.c_int, [C.c_int]),
    "evc_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                  C.POINTER(C.c_int)]),
    "evc_profile_stage": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "evc_profile_select": (C.c_int, [C.c_uint]),
    "evc_profile_kernel": (C.c_char_p, [C.c_int]),
    "evc_release_workspace": (C.c_int, [C.c_void_p]),
}
# stages of evc_profile_stage (include/evcont_hip.h EVC_PROF_*)
PROF_STAGES = {"k5_rows": 0, "k8_cols": 1, "pair_transform": 2, "ip1": 3, "y2": 4, "unpack": 5, "loewdin": 6,
               "subspace": 7}

_lib: Optional[C.CDLL] = None


class EvcontHipError(RuntimeError):
    pass


def load(path: Optional[str] = None) -> C.CDLL:
    """Load the HIP library; raises if it is absent (no silent fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    try:
        # PyTorch-ROCm ships its own HIP runtime: load it FIRST so that this library binds to the same runtime instance
        # (a second, separately initialised libamdhip64 does not see the device torch is driving)
        import torch  # noqa: F401
    except ImportError:
        pass
    p = path or os.environ.get("EVCONT_HIP_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise EvcontHipError(
            f"{p} not found: build it with `python -m evcont_amd.build` (needs hipcc, --offload-arch=gfx950). "
            "evcont_amd has no CPU fallback.")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.evc_abi_version() != ABI_VERSION:
        raise EvcontHipError(f"ABI mismatch: library {lib.evc_abi_version()} != binding {ABI_VERSION}")
    if path is None:
        _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().evc_last_error().decode(errors="replace")
        raise EvcontHipError(f"{what} failed (rc={rc}): {msg}")
