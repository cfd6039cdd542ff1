"""Host-resident geometries, one per call: the MD loop of the reference (``MD_utils.py:40-55``), where PySCF
produces the AO integrals of every step on the host and the continuation consumes them once.

``HostedEvaluator`` keeps, for one molecule size,

* PINNED host staging buffers in the shapes the device path consumes -- with the compressed ``sym8`` training set the
  two large arrays are staged packed, ``int2e`` as PySCF's ``aosym="s4"`` (Ms x Ms) and ``int2e_ip1`` as
  ``aosym="s2kl"`` (3,N,N,Ms): 12 MB instead of 26.6 MB per step at H30.  ``staging()`` hands out numpy views of
  them, so a producer that can write into a caller-supplied array (``mol.intor(..., out=...)``) fills them without an
  extra copy;
* static device buffers, workspace and outputs;
* the step as a two-stream enqueue: upload of the small early inputs (S, hcore, int2e, ...) -> Loewdin, integral
  rotation, H build, eigensolve on the main stream, while a forked stream uploads the late, large inputs
  (``int2e_ip1``, ``dhcore``), which only the gradient tail reads -> join -> predicted RDMs and gradient -> download of
  E and the forces into pinned memory; the PCIe transfer of the 10 MB array overlaps the first half of the device
  work.  DEFAULT: enqueued eagerly every step.  Opt-in (``use_graph=True`` / ``EVCONT_AMD_HOSTED_GRAPH=1``): the same
  enqueue captured once and replayed as ONE HIP graph -- measured slower on MI355X / ROCm 7.2 (0.58 against 0.39 ms at
  H30, no gain for the small systems), kept correct by ``tests/test_gpu_drivers.py``.

No CPU fallback: the step only contains HIP work of ``libevcont_hip.so`` and copies.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .evaluator import BatchedEvaluator, DeviceAOBatch, DeviceTRDMs, F64

_LATE = ("dhcore", "eri_ip1")
_GROUPS = (("S", "hcore", "enuc", "ipovlp", "gnuc", "eri"), _LATE)


def staging_layout(n: int, natm: int, packed: bool):
    """The two staging slabs of a geometry (the early arrays and int2e; dhcore and int2e_ip1): per slab a list of
    (field, shape with a leading batch axis of 1, offset in doubles) and its length.  Every array starts on a 16-byte
    boundary.  ``packed``: int2e as the (Ms, Ms) matrix of ``aosym="s4"``, int2e_ip1 as (3,N,N,Ms) of ``"s2kl"``.
    An integral producer that lays its output out like this in PINNED memory (``synthetic.AOArrays.pinned_packed``)
    is uploaded from there as it stands: two copies per step, no staging copy on the host."""
    npr = n * (n + 1) // 2
    shapes = {"S": (1, n, n), "hcore": (1, n, n), "enuc": (1,), "ipovlp": (1, 3, n, n), "gnuc": (1, natm, 3),
              "dhcore": (1, natm, 3, n, n),
              "eri": (1, npr, npr) if packed else (1, n, n, n, n),
              "eri_ip1": (1, 3, n, n, npr) if packed else (1, 3, n, n, n, n)}
    out = []
    for grp in _GROUPS:
        off, fields = 0, []
        for k in grp:
            size = int(np.prod(shapes[k]))
            fields.append((k, shapes[k], off))
            off += (size + 1) // 2 * 2
        out.append((fields, off))
    return out


class HostedEvaluator:
    def __init__(self, trdms: DeviceTRDMs, natm: int, aoslices, warm_start: bool = True,
                 use_graph: Optional[bool] = None, keep_density_matrices: bool = False,
                 zero_copy: Optional[bool] = None):
        """``use_graph``: replay the step as one HIP graph (default: ``EVCONT_AMD_HOSTED_GRAPH=1``, else off --
        measured on MI355X / ROCm 7.2 the replay of a graph with memcpy nodes is no faster than the eager enqueue for
        the small systems and slower for H30)."""
        if use_graph is None:
            import os
            use_graph = os.environ.get("EVCONT_AMD_HOSTED_GRAPH", "0") not in ("", "0")
        self.t, self.natm = trdms, int(natm)
        d, n = trdms.device, trdms.n
        self.packed = trdms.layout == _lib.LAYOUT_SYM8 and n <= 64
        layout = staging_layout(n, self.natm, self.packed)
        shapes = {k: shp for fields, _ in layout for k, shp, _ in fields}
        # two slabs = two H2D copies per step (a copy costs ~12 us before its first byte moves): the early arrays (the
        # small ones and int2e) and the late pair (dhcore, int2e_ip1); every array is a 16-byte aligned view of its slab
        # SMALL systems (all inputs together below EVCONT_AMD_ZERO_COPY_BYTES, default 1 MiB): no copies at all -- the
        # kernels read the integrals straight from the pinned staging buffers (device-visible on ROCm) and write the
        # energies / forces straight into pinned memory.  An asynchronous copy costs ~10-15 us before its first byte
        # moves and the step has three of them plus two stream joins, which for a 0.13 ms step (H2O 6-31G) is most of the
        # time; reading 0.4 MB through PCIe inside the kernels costs less.
        if zero_copy is None:
            import os
            lim = int(os.environ.get("EVCONT_AMD_ZERO_COPY_BYTES", str(1 << 20)))
            zero_copy = 8 * sum(int(np.prod(v)) for v in shapes.values()) <= lim
        self.zero_copy = bool(zero_copy)
        self._groups = _GROUPS
        self.host: Dict[str, torch.Tensor] = {}
        self.dev: Dict[str, torch.Tensor] = {}
        self._slabs = []
        for fields, total in layout:
            hs = torch.zeros(total, dtype=F64).pin_memory()
            ds = hs if self.zero_copy else torch.zeros(total, dtype=F64, device=d)
            self._slabs.append((hs, ds))
            for k, shp, o in fields:
                x = int(np.prod(shp))
                self.host[k] = hs[o: o + x].view(shp)
                self.dev[k] = ds[o: o + x].view(shp)
        sl = torch.from_numpy(np.ascontiguousarray(np.asarray(aoslices, dtype=np.int64).reshape(-1, 2))).to(d)
        self.aob = DeviceAOBatch(S=self.dev["S"], hcore=self.dev["hcore"], eri=self.dev["eri"], enuc=self.dev["enuc"],
                                 natm=self.natm, ipovlp=self.dev["ipovlp"], dhcore=self.dev["dhcore"],
                                 eri_ip1=self.dev["eri_ip1"], gnuc=self.dev["gnuc"], aoslices=sl,
                                 ip1_s2kl=self.packed, eri_s4=self.packed)
        self.stream = torch.cuda.Stream(d)
        self.side = torch.cuda.Stream(d)
        T = trdms.T
        self._out_slab = torch.zeros(T + max(self.natm, 1) * 3, dtype=F64).pin_memory()
        # (copy mode: energies and gradient come back with one copy, they share a device buffer; letting the kernels
        #  write them into pinned memory instead was measured at 2-4x the step time for H2O / Zundel / H30 -- the
        #  gradient tail reads the buffer back -- so only the no-copy mode of the small systems does that)
        self.ev = BatchedEvaluator(trdms, self.natm, 1, stream=self.stream, warm_start=warm_start,
                                   keep_density_matrices=keep_density_matrices, keep_one_rdm=True,
                                   energy_grad=self._out_slab if self.zero_copy else None)
        self.out_host = {"energy": self._out_slab[:T].view(1, T),
                         "grad": self._out_slab[T:].view(1, max(self.natm, 1), 3)}
        self.use_graph = bool(use_graph)
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        self._calls = 0
        self._sym_checked = False
        # the two large arrays of the staged geometry, when the producer already holds them packed in PINNED memory:
        # uploaded straight from there (no staging copy on the host -- 12 MB per step at H30)
        self._direct: Dict[str, Optional[torch.Tensor]] = {"eri": None, "eri_ip1": None}
        self._direct_slabs = None   # ... or the whole geometry, when the producer laid it out as `staging_layout` does
        self._slab_prefix = [fields[-1][2] for fields, _ in layout]

    # -- staging ------------------------------------------------------------------------------------
    def staging(self) -> Dict[str, np.ndarray]:
        """Pinned numpy views to be filled with the integrals of the next geometry (leading batch axis of 1 removed;
        ``enuc`` has shape (1,)).  ``eri`` / ``eri_ip1`` are in the packed forms when ``self.packed``."""
        return {k: (v.numpy()[0] if k != "enuc" else v.numpy()) for k, v in self.host.items()}

    def stage(self, ao) -> None:
        """Copy an ``AOArrays``-like object (numpy fields) into the staging buffers, packing the two large arrays on
        the host when the device side wants them packed and they arrive full."""
        st = self.staging()
        n = self.t.n
        self._direct_slabs = None
        slabs = getattr(ao, "_staging_slabs", None)
        if (slabs is not None and not self.zero_copy and getattr(ao, "_staging_key", None) == (n, self.natm, self.packed)
                and all(a.numel() == b[0].numel() and a.is_pinned() for a, b in zip(slabs, self._slabs))):
            if self.packed and not self._sym_checked:
                from .evaluator import _CHECK_SYM, check_integral_symmetry
                if _CHECK_SYM != "0" and not getattr(ao, "integral_symmetry", False):
                    check_integral_symmetry(np.asarray(ao.eri), np.asarray(ao.eri_ip1), n, what="HostedEvaluator.stage")
                self._sym_checked = True
            self._direct_slabs = slabs      # the producer's two pinned slabs ARE the staging buffers of this step
            self._direct = {"eri": None, "eri_ip1": None}
            return
        for k in ("S", "hcore", "ipovlp", "dhcore", "gnuc"):
            np.copyto(st[k], np.asarray(getattr(ao, k), dtype=np.float64).reshape(st[k].shape))
        st["enuc"][0] = float(ao.enuc)
        eri, ip1 = np.asarray(ao.eri), np.asarray(ao.eri_ip1)
        if self.packed:
            from .evaluator import _CHECK_SYM, check_integral_symmetry, spot_check_integral_symmetry
            # (packing below keeps one triangle: it must be THE triangle; complete check once per evaluator -- with
            #  EVCONT_AMD_CHECK_SYM=2 every call --, a random sample on every other call, unless the molecule declares
            #  the symmetry itself)
            if not getattr(ao, "integral_symmetry", False) and _CHECK_SYM != "0":
                if _CHECK_SYM == "2" or not self._sym_checked:
                    check_integral_symmetry(eri, ip1, n, what="HostedEvaluator.stage")
                elif eri.size != st["eri"].size or ip1.size != st["eri_ip1"].size:   # (full arrays about to be packed)
                    spot_check_integral_symmetry(eri, ip1, n, samples=1024,
                                                 what="HostedEvaluator.stage")
            self._sym_checked = True
            iu, ju = np.tril_indices(n)
            if eri.size != st["eri"].size:
                eri = eri.reshape(n, n, n, n)[iu, ju][:, iu, ju]
            if ip1.size != st["eri_ip1"].size:
                ip1 = ip1.reshape(3, n, n, n, n)[:, :, :, iu, ju]
        for k, src in (("eri", eri), ("eri_ip1", ip1)):
            self._direct[k] = None
            if (not self.zero_copy and src.size == st[k].size and src.dtype == np.float64 and src.flags.c_contiguous
                    and src.flags.writeable and src.nbytes >= (1 << 20)):
                tsrc = torch.from_numpy(src.reshape(-1))
                if tsrc.is_pinned():          # the producer's own pinned buffer: uploaded from there
                    self._direct[k] = tsrc
                    continue
            np.copyto(st[k], src.reshape(st[k].shape))

    # -- one step -----------------------------------------------------------------------------------
    def _enqueue_step(self) -> None:
        """Uploads, the device DAG and the downloads, on self.stream with self.side forked for the late inputs."""
        main, side = self.stream, self.side
        if self.zero_copy:   # the kernels read the pinned buffers themselves and write the results into pinned memory
            with torch.cuda.stream(main):
                self.ev.enqueue(self.aob, 1, energy_only=False)
            return
        (h0, d0), (h2, d2) = self._slabs
        # Early slab on the main stream, the late slab on a forked one, joined in front of the gradient tail.
        # (Measured alternatives on MI355X / ROCm 7.2, H30, per step: this order 425 us -- 393-402 us since the small
        #  arrays and int2e travel in ONE copy and energies + gradient come back in one --; int2e on the forked stream as
        #  well with the Loewdin kernel started behind the small slab alone, evc_phase_loewdin_batch: 530-540 us -- every
        #  cross-stream event wait on the critical path costs 20-25 us and the runtime takes tens of microseconds on
        #  the host to accept a 10 MB copy; the same with the Loewdin kernel reading S / hcore straight from the pinned
        #  host buffers (no upload in front of it at all) and every copy on the forked stream: 430-515 us; the same
        #  enqueue replayed as a HIP graph: 555-590 us.)
        p0, p2 = self._slab_prefix
        e_src, i_src = self._direct["eri"], self._direct["eri_ip1"]
        if self._direct_slabs is not None:
            h0, h2 = self._direct_slabs
        with torch.cuda.stream(main):                     # submitted FIRST: the copy engine works in submission order
            if e_src is None:
                d0.copy_(h0, non_blocking=True)
            else:                                         # small arrays from the slab, int2e from the producer's buffer
                d0[:p0].copy_(h0[:p0], non_blocking=True)
                d0[p0: p0 + e_src.numel()].copy_(e_src, non_blocking=True)
        side.wait_stream(main)                            # fork (behind the early copy)
        with torch.cuda.stream(side):
            if i_src is None:
                d2.copy_(h2, non_blocking=True)
            else:
                d2[:p2].copy_(h2[:p2], non_blocking=True)
                d2[p2: p2 + i_src.numel()].copy_(i_src, non_blocking=True)
        with torch.cuda.stream(main):
            self.ev.enqueue(self.aob, 1, energy_only=True)       # Loewdin .. eigensolve (what the tail needs stays in the workspace)
            main.wait_stream(side)                        # join: the gradient tail reads eri_ip1 and dhcore
            self.ev.phase_gradient(self.aob, False)
            self._out_slab.copy_(self.ev.energy_grad, non_blocking=True)

    def run(self):
        """Evaluate the geometry in the staging buffers: ``(E_total, grad (A,3))`` as numpy."""
        self._calls += 1
        # the first two calls run eagerly: the first primes the warm start (so the flag captured below is the one
        # every later step uses) and performs the one-time set-up of the library; the third call captures
        if not self.use_graph or self._calls <= 2:
            self._enqueue_step()
        else:
            if self.graph is None:
                self.stream.synchronize()
                self.side.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=self.stream):
                    self._enqueue_step()
                self.graph = g
            with torch.cuda.stream(self.stream):
                self.graph.replay()
        self.stream.synchronize()
        e = float(self.out_host["energy"][0, 0])
        if not np.isfinite(e):
            raise np.linalg.LinAlgError("generalised eigenproblem failed: overlap matrix not positive definite "
                                        "or non-finite input")
        return e, self.out_host["grad"][0, : self.natm].numpy().copy()

    def predicted_one_rdm(self) -> np.ndarray:
        """Predicted 1-RDM (OAO basis) of the last evaluated geometry."""
        return self.ev.d_pred[0].cpu().numpy().copy()

    def energy_with_grad(self, ao, return_density_matrices: bool = False):
        self.stage(ao)
        e, g = self.run()
        if return_density_matrices:
            return e, g, self.ev.d_pred[0].cpu().numpy().copy(), self.ev.g_pred[0].cpu().numpy().copy()
        return e, g
