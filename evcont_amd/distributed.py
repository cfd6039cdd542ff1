"""Pair-sharded multi-GPU evaluation: one process per GPU, each rank holds a
contiguous slice of the two-body t-RDM rows (training pairs).  Per geometry:

    phase A (local)   Loewdin + integrals (replicated, cheap) + H rows of the local pairs
    all_gather        the scaled H rows           -- P doubles in total (KB-sized)
    phase B (local)   identical T x T eigensolve on every rank (no broadcast of c needed)
    phase C (local)   partial predicted 2-RDM of the local pairs pushed through the gradient,
                      which is LINEAR in it (``ab_initio_gradients_loewdin.py:210-252,300-303``);
                      rank 0 alone adds the one-body and nuclear terms
    all_reduce(SUM)   the (A,3) gradient          -- 720 B at A=30

``torch.distributed`` with backend "nccl" is RCCL over xGMI on ROCm; the same code runs on
"gloo" for the CPU tests, where the three phases are supplied by a test double.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_rows(rows_total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [r0, r1) of rank `rank`; all slices but the last have ceil(rows/world) rows
    so that the all-gather can use one fixed chunk size."""
    chunk = -(-rows_total // world)
    r0 = min(rows_total, rank * chunk)
    return r0, min(rows_total, r0 + chunk)


class PairShardedContinuation:
    """Drives a per-rank evaluator (``evaluator.ContinuationEvaluator`` on its row slice, or any
    object with the same three phase methods and ``grad``/``energy`` tensors) and the two collectives."""

    def __init__(self, evaluator, rows_total: int, group: Optional[dist.ProcessGroup] = None):
        self.ev = evaluator
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.rows_total = int(rows_total)
        self.chunk = -(-self.rows_total // self.world)
        self.r0, self.r1 = shard_rows(self.rows_total, self.world, self.rank)
        dev = evaluator.grad.device
        self._send = torch.zeros(self.chunk, dtype=torch.float64, device=dev)
        self._recv = torch.zeros(self.chunk * self.world, dtype=torch.float64, device=dev)

    def enqueue(self, ao, nroots: int = 1, energy_only: bool = False) -> None:
        rows_local = self.ev.phase_hamiltonian(ao)
        n_local = self.r1 - self.r0
        if n_local:
            self._send[:n_local].copy_(rows_local[:n_local])
        dist.all_gather_into_tensor(self._recv, self._send, group=self.group)
        rows_all = self._recv[: self.rows_total]      # chunks are contiguous and only the tail is padding
        self.ev.phase_solve(ao, rows_all.contiguous(), nroots)
        if energy_only:
            return
        self.ev.phase_gradient(ao, partial_rank=(self.rank != 0))
        dist.all_reduce(self.ev.grad, op=dist.ReduceOp.SUM, group=self.group)

    def energy_with_grad(self, ao):
        self.enqueue(ao)
        if self.ev.grad.is_cuda:
            torch.cuda.current_stream(self.ev.grad.device).synchronize()
        return float(self.ev.energy[0].item()), self.ev.grad.cpu().numpy().copy()
