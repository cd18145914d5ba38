"""The N > 1 path on CPU: world_size-2 gloo processes shard a read set, count their shards (here
with the oracle standing in for the GPU kernels -- this test is about sharding + the reduce step),
all-reduce the count vectors through screencounter_amd.parallel and must reproduce the
single-process result exactly."""
import os
import random
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle.pyoracle import Oracle
    from screencounter_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = parallel.shard_bounds(len(case["reads"]), rank, world)
        counts, total = Oracle().count_single(case["reads"][lo:hi], case["template"], case["strand"], case["pool"],
                                              case["mismatches"], case["use_first"])
        t = torch.from_numpy(counts.astype(np.int32))
        t, total = parallel.all_reduce_counts(t, total)
        files = parallel.assign_files(5, rank, world)
        cols = parallel.gather_columns({i: i * 10 for i in files}, 5)
        if rank == 0:
            np.save(out_path, np.concatenate([t.numpy().astype(np.int64), [total], cols]))
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    from screencounter_amd import parallel
    for n in (0, 1, 7, 100, 101):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.assign_files(5, 1, 2) == [1, 3]


def test_two_rank_gloo_reduce_matches_single_process(tmp_path, oracle):
    import torch.multiprocessing as mp
    from tests import gen
    rng = random.Random(77)
    case = gen.random_single_case(rng, max_vlen=10, sizes=(200,))
    while len(case["pool"]) < 2 or len(case["reads"]) < 50:
        case = gen.random_single_case(rng, max_vlen=10, sizes=(200,))
    try:
        exp_counts, exp_total = oracle.count_single(case["reads"], case["template"], case["strand"], case["pool"],
                                                    case["mismatches"], case["use_first"])
    except Exception:
        pytest.skip("degenerate random case")
    out = str(tmp_path / "out.npy")
    mp.spawn(_worker, args=(2, _free_port(), case, out), nprocs=2, join=True)
    got = np.load(out)
    n = len(case["pool"])
    assert np.array_equal(got[:n], exp_counts.astype(np.int64))
    assert got[n] == exp_total == len(case["reads"])
    assert got[n + 1:].tolist() == [0, 10, 20, 30, 40]
