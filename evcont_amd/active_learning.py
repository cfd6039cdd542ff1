"""Device-side pieces of the active-learning MD loop (``MD_utils.converge_EVCont_MD``,
``MD_utils.py:128-502``) and the loop itself for molecules that can be rebuilt at new
coordinates without PySCF.

After every trajectory the reference (i) re-evaluates the continuation energy along it with the
training set minus its newest state — and, when pruning, minus each state in turn: ``steps x (T+1)``
full evaluations, each contracting the whole t-RDM (``:264-299,448-483``) — and (ii) picks the next
training geometry, by default the trajectory point whose OAO integrals are farthest from all training
integrals, ``min_j |h1-h1_j|^2 + 1/2 |h2-h2_j|^2`` (``:363-405``), rotating the integrals of every
point once more.

Here (i) costs ONE batched pass per 32 geometries: the subspace matrix of a subset of the training
states is the corresponding sub-matrix of the full ``H(R)``, so the t-RDM is contracted once per
geometry (``evc_energy_with_grad_batch`` with ``hmat`` kept) and all ``T+1`` subset problems are small
generalised eigenproblems solved as one batched launch (``evc_subspace_solve_batch``); (ii) is one
batched Loewdin + four-index transform over the trajectory (``evc_integrals_oao_batch``) and plain
device reductions.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import check
from .evaluator import BatchedEvaluator, DeviceAO, DeviceAOBatch, DeviceTRDMs, F64, _dev, _stream_ptr

MAX_BATCH = 64      # geometries per batched call (bounds the workspace: ~30 MB each at H30)


# ------------------------------------------------------------------------------------------------
# building blocks
# ------------------------------------------------------------------------------------------------
def integrals_oao_batch(S: torch.Tensor, hcore: torch.Tensor, eri: torch.Tensor):
    """(h1 (B,N,N), h2 (B,N,N,N,N), X (B,N,N)) of B geometries on the device."""
    lib = _lib.load()
    B, n = int(S.shape[0]), int(S.shape[1])
    d = S.device
    h1 = torch.empty((B, n, n), dtype=F64, device=d)
    h2 = torch.empty((B, n, n, n, n), dtype=F64, device=d)
    X = torch.empty((B, n, n), dtype=F64, device=d)
    nbytes = lib.evc_integrals_oao_ws_bytes(n, B)
    if nbytes == 0:
        raise _lib.EvcontHipError("evc_integrals_oao_ws_bytes: n or count out of range")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=d)
    S, hcore, eri = S.contiguous(), hcore.contiguous(), eri.contiguous()
    check(lib.evc_integrals_oao_batch(n, B, S.data_ptr(), hcore.data_ptr(), eri.data_ptr(), h1.data_ptr(),
                                      h2.data_ptr(), X.data_ptr(), ws.data_ptr(), nbytes, _stream_ptr(d)),
          "evc_integrals_oao_batch")
    return h1, h2, X


def subspace_energies(H: torch.Tensor, S: torch.Tensor, e_shift: Optional[torch.Tensor] = None,
                      nroots: int = 1) -> torch.Tensor:
    """Lowest ``nroots`` generalised eigenvalues of ``count`` problems ``H[g] c = E S[g] c`` (lower triangles,
    ``scipy.linalg.eigh(H, S)`` semantics) -> (count, nroots).  ``S`` is (T,T) (shared) or (count,T,T)."""
    lib = _lib.load()
    assert H.dim() == 3 and H.shape[1] == H.shape[2] and H.dtype == F64
    count, T = int(H.shape[0]), int(H.shape[1])
    H = H.contiguous()
    S = S.contiguous()
    shared = S.dim() == 2
    assert tuple(S.shape[-2:]) == (T, T) and (shared or S.shape[0] == count)
    evals = torch.empty((count, T), dtype=F64, device=H.device)
    evecs = torch.empty((count, T, T), dtype=F64, device=H.device)
    es = e_shift.contiguous() if e_shift is not None else None
    # T > 32: the kernel needs scratch per problem (evc_subspace_solve_ws_bytes); many problems go in chunks of <= 1 GiB
    per = lib.evc_subspace_solve_ws_bytes(T, 1)
    chunk = count if per == 0 else max(1, min(count, (1 << 30) // per))
    ws = torch.empty(per * chunk, dtype=torch.uint8, device=H.device) if per else None
    for c0 in range(0, count, chunk):
        c1 = min(count, c0 + chunk)
        check(lib.evc_subspace_solve_batch(H[c0:c1].data_ptr(), S.data_ptr() if shared else S[c0:c1].data_ptr(),
                                           0 if shared else T * T, T, c1 - c0, int(nroots),
                                           es[c0:c1].data_ptr() if es is not None else None, evals[c0:c1].data_ptr(),
                                           evecs[c0:c1].data_ptr(), ws.data_ptr() if ws is not None else None,
                                           per * (c1 - c0), _stream_ptr(H.device)), "evc_subspace_solve_batch")
    return evals[:, :nroots]


def trajectory_hamiltonians(trd: DeviceTRDMs, aos: Sequence[DeviceAO]):
    """(H (B,T,T), E (B,), enuc (B,)) along a list of geometries: batched energy-only evaluations with the
    subspace matrices kept (lower triangles, exactly what the reference hands to ``eigh``)."""
    Hs, Es, en = [], [], []
    natm = max(1, aos[0].natm)
    evs = {}
    for k in range(0, len(aos), MAX_BATCH):
        chunk = list(aos[k:k + MAX_BATCH])
        G = len(chunk)
        if G not in evs:
            evs[G] = BatchedEvaluator(trd, natm, G, keep_hmat=True)
        be = evs[G]
        aob = DeviceAOBatch.stack(chunk)
        be.enqueue(aob, nroots=1, energy_only=True)
        be.synchronize()
        Hs.append(be.hmat.clone())
        Es.append(be.energy[:, 0].clone())
        en.append(aob.enuc.clone())
    return torch.cat(Hs), torch.cat(Es), torch.cat(en)


def subset_energies(H: torch.Tensor, S: torch.Tensor, enuc: torch.Tensor, subsets: Sequence[Sequence[int]]):
    """Continuation energies (B, len(subsets)) when only the training states ``subsets[k]`` are used
    (``one_rdm[np.ix_(ids, ids)]`` in the reference, ``MD_utils.py:264-299``).  Subsets of equal size are
    solved together."""
    B = int(H.shape[0])
    out = torch.empty((B, len(subsets)), dtype=F64, device=H.device)
    by_size = {}
    for k, ids in enumerate(subsets):
        by_size.setdefault(len(ids), []).append(k)
    for size, ks in by_size.items():
        idx = torch.tensor([list(subsets[k]) for k in ks], dtype=torch.long, device=H.device)   # (K,size)
        Hs = H[:, idx[:, :, None], idx[:, None, :]]                                              # (B,K,size,size)
        Ss = S[idx[:, :, None], idx[:, None, :]]                                                 # (K,size,size)
        K = len(ks)
        e = subspace_energies(Hs.reshape(B * K, size, size), Ss.unsqueeze(0).expand(B, K, size, size)
                              .reshape(B * K, size, size), enuc.repeat_interleave(K))
        out[:, ks] = e.reshape(B, K)
    return out


def hamiltonian_distances(h1: torch.Tensor, h2: torch.Tensor, h1_trn: torch.Tensor, h2_trn: torch.Tensor):
    """d[b, j] = |h1[b]-h1_trn[j]|^2 + 1/2 |h2[b]-h2_trn[j]|^2  (``MD_utils.py:393-396``)."""
    B, T = h1.shape[0], h1_trn.shape[0]
    d = torch.empty((B, T), dtype=F64, device=h1.device)
    a1, a2 = h1.reshape(B, -1), h2.reshape(B, -1)
    for j in range(T):
        d[:, j] = ((a1 - h1_trn[j].reshape(1, -1)) ** 2).sum(dim=1) + 0.5 * ((a2 - h2_trn[j].reshape(1, -1)) ** 2).sum(dim=1)
    return d


def _stack_integral_inputs(mols, device):
    up = lambda name: torch.from_numpy(np.ascontiguousarray(np.stack([np.asarray(getattr(m, name), dtype=np.float64)
                                                                      for m in mols]))).to(device)
    return up("S"), up("hcore"), up("eri")


def farthest_point_ham(traj_mols, trn_mols, device=None) -> int:
    """Index of the trajectory geometry whose OAO integrals are farthest from those of all training
    geometries (first maximum, like the reference's strict ``>`` scan, ``MD_utils.py:397-400``)."""
    d = _dev(device)
    h1t, h2t, _ = integrals_oao_batch(*_stack_integral_inputs(trn_mols, d))
    best, best_val = 0, None
    for k in range(0, len(traj_mols), MAX_BATCH):
        h1, h2, _ = integrals_oao_batch(*_stack_integral_inputs(traj_mols[k:k + MAX_BATCH], d))
        m = hamiltonian_distances(h1, h2, h1t, h2t).min(dim=1).values.cpu().numpy()
        j = int(np.argmax(m))
        if best_val is None or m[j] > best_val:
            best, best_val = k + j, float(m[j])
    return best


# ------------------------------------------------------------------------------------------------
# the loop
# ------------------------------------------------------------------------------------------------
def _device_aos(mols, device, energy_only=True) -> List[DeviceAO]:
    return [DeviceAO.from_arrays(m, device, energy_only=energy_only) for m in mols]


def converge_EVCont_MD(EVCont_obj, init_mol, steps=100, dt=1, convergence_thresh=1.0e-3,
                       prune_irrelevant_data=False, trn_times=None, data_addition="farthest_point_ham",
                       max_iterations: Optional[int] = None, workdir: str = "."):
    """On-the-fly training of a continuation along its own MD trajectory (``MD_utils.py:128-502``).

    Iteration ``i``: propagate ``steps`` NVE steps from ``init_mol`` with the current training set (files
    ``traj_EVCont_i.{xyz,npy}``, ``ens_EVCont_i.xyz``), compare the energies along the trajectory with those
    the previous training set (newest state removed) gives on the same geometries (``en_diff_i.txt``); stop
    after two consecutive iterations below ``convergence_thresh``; otherwise add the trajectory point
    selected by ``data_addition`` ("farthest_point_ham", "farthest_point" or "energy") as a training state
    (``overlap/one_rdm/two_rdm.npy``, ``trn_times.txt``) and repeat.  With ``prune_irrelevant_data`` states
    whose removal changes no energy of the trajectory by ``convergence_thresh`` or more are dropped.

    ``init_mol`` must offer ``with_coords(coords)`` (e.g. ``hchain.HChainMol``); PySCF molecules are served
    by the reference's own driver on top of this package's evaluator functions.  Restarting from files
    (``trn_times`` given) is not supported.  Returns the last trajectory ``(steps, A, 3)``.
    """
    from .MD_utils import get_trajectory
    if trn_times:
        raise NotImplementedError("converge_EVCont_MD: restart from a previous run (trn_times) is not supported")
    if not hasattr(init_mol, "with_coords"):
        raise NotImplementedError("converge_EVCont_MD needs a molecule with with_coords(); use the reference's "
                                  "driver for PySCF molecules")
    assert data_addition in ("farthest_point_ham", "farthest_point", "energy")
    path = lambda name: os.path.join(workdir, name)
    dev = _dev()
    trn_times, trn_geoms = [0], [np.array(init_mol.atom_coords())]
    EVCont_obj.append_to_rdms(init_mol)

    def save_training(i):
        suffix = f"_{i}" if prune_irrelevant_data else ""
        np.save(path(f"overlap{suffix}.npy"), EVCont_obj.overlap)
        np.save(path(f"one_rdm{suffix}.npy"), EVCont_obj.one_rdm)
        np.save(path(f"two_rdm{suffix}.npy"), EVCont_obj.two_rdm)
        if i > 0:
            np.savetxt(path(f"trn_times{suffix}.txt"), np.array(trn_times))

    def run_trajectory(i):
        traj = get_trajectory(init_mol, EVCont_obj.overlap, EVCont_obj.one_rdm, EVCont_obj.two_rdm, steps=steps, dt=dt,
                              trajectory_output=path(f"traj_EVCont_{i}.xyz"), energy_output=path(f"ens_EVCont_{i}.xyz"))
        np.save(path(f"traj_EVCont_{i}.npy"), traj)
        ens = np.atleast_2d(np.genfromtxt(path(f"ens_EVCont_{i}.xyz")))[:, 1]
        return traj, np.ascontiguousarray(ens)

    i = 0
    save_training(i)
    trajectory, updated_ens = run_trajectory(i)
    reference_ens = updated_ens[0]
    converged = False
    while True:
        en_diff = np.abs(reference_ens - updated_ens)
        np.savetxt(path(f"en_diff_{i}.txt"), np.atleast_1d(en_diff))
        i += 1
        if converged and en_diff.max() <= convergence_thresh:
            break
        converged = bool(en_diff.max() <= convergence_thresh)
        if max_iterations is not None and i > max_iterations:
            break
        traj_mols = [init_mol.with_coords(g, need_grad=False) for g in trajectory]
        if data_addition == "energy":
            trn_time = int(np.argmax(en_diff))
        elif data_addition == "farthest_point":
            d2 = np.array([np.sum(np.abs(g - trajectory) ** 2, axis=(-1, -2)) for g in trn_geoms])
            trn_time = int(np.argmax(np.min(d2, axis=0)))
        else:
            trn_time = farthest_point_ham(traj_mols, [init_mol.with_coords(g, need_grad=False) for g in trn_geoms], dev)
        trn_times.append(trn_time)
        trn_geoms.append(np.array(trajectory[trn_time]))
        EVCont_obj.append_to_rdms(init_mol.with_coords(trajectory[trn_time], need_grad=False))
        save_training(i)
        trajectory, updated_ens = run_trajectory(i)
        # energies of the new trajectory with subsets of the training set: ONE t-RDM contraction per geometry
        T = EVCont_obj.ntrain
        traj_mols = [init_mol.with_coords(g, need_grad=False) for g in trajectory]
        from .ab_initio_eigenvector_continuation import get_trdm_compression, integrals_have_symmetry
        mode = get_trdm_compression()
        lay = "sym8" if mode == "sym8" or (mode == "auto" and integrals_have_symmetry(traj_mols[0])) else "pack2"
        H, _, enuc = trajectory_hamiltonians(EVCont_obj.device_trdms(lay, device=dev), _device_aos(traj_mols, dev))
        S_dev = torch.from_numpy(np.ascontiguousarray(EVCont_obj.overlap, dtype=np.float64)).to(dev)
        reference_ens = subset_energies(H, S_dev, enuc, [list(range(T - 1))])[:, 0].cpu().numpy()
        if prune_irrelevant_data:
            keep = np.ones(T, dtype=bool)
            for j in range(T):
                test = keep.copy()
                test[j] = False
                if test.sum() >= 1:
                    e_removed = subset_energies(H, S_dev, enuc, [list(np.nonzero(test)[0])])[:, 0].cpu().numpy()
                    if np.all(np.abs(e_removed - updated_ens) < convergence_thresh):
                        keep = test
            keep_ids = [int(k) for k in np.nonzero(keep)[0]]
            trn_times = [trn_times[k] for k in keep_ids]
            trn_geoms = [trn_geoms[k] for k in keep_ids]
            EVCont_obj.prune_datapoints(keep_ids)
    return trajectory
