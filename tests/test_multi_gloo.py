"""CPU, world_size 2 over gloo: the multi-GPU merge of per-shard syncmer tables (oatk_amd/multi.py) reproduces the
single-process count -- same global key array, same coverage -- when reads are sharded by record."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import adversarial as A
import oracle_lib as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, reads, K, S, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oatk_amd.multi import merge_numpy
    per = (len(reads) + world - 1) // world
    mine = reads[rank * per:(rank + 1) * per]
    _, c = O.scan_and_count(mine, K, S, mode=1)
    G, Sm, C, l2g = merge_numpy(c["h"], c["s"], c["cov"], dist)
    assert np.array_equal(G[l2g], c["h"])
    np.savez(os.path.join(outdir, "r%d.npz" % rank), G=G, S=Sm, C=C)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("K,S", [(101, 11), (1001, 31)])
def test_sharded_merge_matches_single_process(tmp_path, K, S):
    reads = A.hifi_like(60, 8000 if K < 1001 else 40000, 1500 if K < 1001 else 9000, seed=K + 1)
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), reads, K, S, str(tmp_path)), nprocs=world, join=True)
    _, full = O.scan_and_count(reads, K, S, mode=1)
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert np.array_equal(z["G"], full["h"])          # global ids = ranks in the merged key array
        assert np.array_equal(z["S"], full["s"])
        assert np.array_equal(z["C"], full["cov"].astype(np.int64))


def test_single_rank_merge_is_identity():
    from oatk_amd.multi import merge_numpy
    reads = A.hifi_like(20, 5000, 1200, seed=4)
    _, c = O.scan_and_count(reads, 101, 11, mode=1)
    G, Sm, C, l2g = merge_numpy(c["h"], c["s"], c["cov"])
    assert np.array_equal(G, c["h"]) and np.array_equal(C, c["cov"].astype(np.int64)) and np.array_equal(l2g, np.arange(len(G)))


def _gather_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oatk_amd.multi import all_reduce, gather_var
    # ragged first dimension, fixed trailing dimension: the shapes ShardedEc moves (pair lists, k-mer strings, id lists)
    a = torch.arange(3 + 4 * rank, dtype=torch.int64) + 100 * rank
    b = (torch.arange((rank * 2) * 16, dtype=torch.int32) % 251).to(torch.uint8).reshape(rank * 2, 16)      # rank 0 contributes nothing
    ga, gb = gather_var(a, dist), gather_var(b, dist)
    assert [t.shape[0] for t in ga] == [3 + 4 * r for r in range(world)] and [t.shape for t in gb] == [(2 * r, 16) for r in range(world)]
    assert torch.equal(ga[rank], a) and torch.equal(gb[rank], b)
    assert torch.equal(torch.cat(ga), torch.cat([torch.arange(3 + 4 * r, dtype=torch.int64) + 100 * r for r in range(world)]))
    owner = all_reduce(torch.tensor([rank if rank else world, 7 - rank]), dist, dist.ReduceOp.MIN)
    assert owner.tolist() == [1, 7 - (world - 1)]
    tot = all_reduce(torch.tensor([rank + 1], dtype=torch.int64), dist)
    assert int(tot.item()) == world * (world + 1) // 2
    dist.barrier()
    dist.destroy_process_group()


def test_ragged_gather_and_reductions_of_the_sharded_ec_round():
    """the collectives ShardedEc is made of (oatk_amd/multi.py), world_size 3 over gloo; the device calls between them are
    covered on the GPU by tests/test_gpu_sharded_ec.py"""
    world = 3
    mp.spawn(_gather_worker, args=(world, _free_port(), ""), nprocs=world, join=True)
