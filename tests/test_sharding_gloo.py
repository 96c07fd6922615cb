"""N>1 host logic on CPU: world_size-2 gloo.  Each rank processes its utterance slice (with the
CPU oracle standing in for the GPU op -- test infrastructure), rank 0 gathers; the result must
equal the single-process result, and the job time is the max over ranks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wavernn_amd import sharding, synth


def test_partition_properties():
    for n in (0, 1, 7, 64, 1000):
        for w in (1, 2, 3, 8):
            parts = sharding.partition(n, w)
            assert len(parts) == w and parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.partition(4, 0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import taco_oracle as O
        x, dur = synth.lr_inputs(6, 9, 8, seed=4, dur_lo=2, dur_hi=2)   # equal lengths: shards concatenate
        lo, hi = sharding.my_slice(x.shape[0], rank, world)
        out = O.length_regulate(torch.as_tensor(x[lo:hi]), torch.as_tensor(dur[lo:hi].copy()))
        dist.barrier()
        t = sharding.max_over_ranks(0.5 + rank)
        full = sharding.gather_to_rank0(out)
        if rank == 0:
            q.put((t, full.numpy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    t, full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle import taco_oracle as O
    x, dur = synth.lr_inputs(6, 9, 8, seed=4, dur_lo=2, dur_hi=2)
    ref = O.length_regulate(torch.as_tensor(x), torch.as_tensor(dur.copy())).numpy()
    assert t == 1.5                       # max over ranks
    assert np.array_equal(full, ref)
