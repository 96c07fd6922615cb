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
        # the same through an explicit host group, as bench.py passes it next to an RCCL default group: a second gloo
        # group over the same ranks, made by dist.new_group
        assert sharding.host_group() is dist.group.WORLD      # the default group is gloo already
        hg = dist.new_group(backend="gloo")
        t2 = sharding.max_over_ranks(0.25 + 2 * rank, device=torch.device("cpu"), group=hg)
        full2 = sharding.gather_to_rank0(out, group=hg)
        wavs = sharding.gather_list_to_rank0([np.full(3 + rank, float(rank))], group=hg)
        if rank == 0:
            assert t2 == 2.25 and torch.equal(full, full2)
            assert [w.tolist() for w in wavs] == [[0.0] * 3, [1.0] * 4]
            q.put((t, full.numpy()))
        else:
            assert full2 is None and wavs is None
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


def test_bench_rejects_mismatched_launcher():
    """`--gpus` must agree with the launcher's WORLD_SIZE (a silent one-GPU run reporting n_gpus: 1 is what this guards)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=4" in p.stderr


@pytest.mark.gpu
def test_bench_gpus_2_spawns_two_ranks():
    """`python bench.py --gpus 2` with no launcher starts two rank processes itself and rank 0 prints ONE JSON line with
    n_gpus == 2.  Rehearsed on the one GPU of the test box (both ranks on cuda:0, gloo instead of RCCL, which refuses two
    ranks on one device); the configs[3] sharded run goes along with 6 utterances."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WRNN_BENCH_REHEARSE_ONE_GPU="1", WRNN_BENCH_E2E_UTTS="6")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert len(out["extras"]["per_rank_frames_per_sec"]) == 2
    e2e = out["extras"]["end_to_end_1k_utterances"]
    assert e2e["utterances"] == 6 and e2e["samples_per_sec"] > 0
