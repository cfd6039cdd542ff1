"""The pair-sharded evaluation through RCCL itself: a FRESH child process (one rank, WORLD_SIZE=1, backend "nccl")
drives the real HIP phases with ``all_gather_into_tensor`` and ``all_reduce`` between them
(evcont_amd/distributed.py) and compares with the CPU oracle.  More RCCL ranks need more GPUs than
this box has; TWO ranks sharing the card on gloo drive the real phases through ``PipelinedPairSharded`` below
(tests/gloo2_child.py); the sharding arithmetic for world sizes 2 and 3 is covered on gloo with a test double
(tests/test_distributed_gloo.py) and with the real phases on emulated ranks (tests/test_gpu_batch.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_pair_sharded_phases_through_rccl_world1():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_child.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + "\n" + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_CHILD ")][-1]
    res = json.loads(line[len("RCCL_CHILD "):])
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["worst_dE"] < 1e-10 and res["worst_dgrad"] < 1e-9, res


def test_pipelined_pair_sharded_two_ranks():
    """``distributed.PipelinedPairSharded`` with TWO ranks (both on the one card, backend gloo): several pair-sharded
    batches in flight on internal streams, their all-gathers and all-reduces issued in program order on one communicator
    by both ranks, different geometries in every slot; and the predicted RDMs summed over the ranks."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "gloo2_child.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    for p, (so_, se_) in zip(procs, outs):
        assert p.returncode == 0, so_[-2000:] + "\n" + se_[-4000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("GLOO2_CHILD ")][-1]
    res = json.loads(line[len("GLOO2_CHILD "):])
    assert res["world"] == 2
    assert res["worst_dE"] < 1e-10 and res["worst_dgrad"] < 1e-9 and res["worst_drdm"] < 1e-10, res
