"""CPU, world_size 2 over gloo: the multi-GPU stream-sharding helpers (audiodec_amd/shard.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from audiodec_amd import shard


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard.stream_range(5)
        assert (lo, hi) == ((0, 3) if rank == 0 else (3, 5))
        # rank 0's weights reach everyone bit for bit
        g = torch.Generator().manual_seed(7 + rank)
        sd = {"a.weight": torch.randn(4, 3, generator=g), "b.bias": torch.randn(5, generator=g)}
        out = shard.broadcast_state_dict(sd, src=0)
        g0 = torch.Generator().manual_seed(7)
        assert torch.equal(out["a.weight"], torch.randn(4, 3, generator=g0)) and torch.equal(out["b.bias"], torch.randn(5, generator=g0))
        # per-rank codes (n_q, B_local, T) gather to (n_q, B_total, T) on rank 0 in stream order
        idx = torch.arange(8 * (hi - lo) * 2).reshape(8, hi - lo, 2) + 1000 * rank
        allc = shard.gather_codes(idx, dst=0)
        if rank == 0:
            assert allc.shape == (8, 5, 2)
            assert torch.equal(allc[:, :3], idx) and int(allc[0, 3, 0]) == 1000
        else:
            assert allc is None
        assert shard.max_over_ranks(1.0 + rank, "cpu") == 2.0
        assert shard.gather_floats(10.0 + rank, "cpu") == [10.0, 11.0]         # every rank's own rate, on every rank (bench.py)
        q.put((rank, "ok"))
    except Exception as e:          # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_stream_sharding_over_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == {0: "ok", 1: "ok"}, res


def test_stream_range_partitions_everything():
    for n, w in ((2048, 8), (5, 2), (7, 3), (1, 4)):
        seen = []
        for r in range(w):
            lo, hi = shard.stream_range(n, r, w)
            seen += list(range(lo, hi))
            for s in range(lo, hi):
                assert shard.owner_of(s, n, w) == r
        assert seen == list(range(n))


def test_rank_cpus_are_disjoint_equal_blocks():
    """shard.rank_cpus: N ranks of one host get disjoint, equally sized, contiguous blocks of the cpus the launcher may use."""
    cpus = list(range(3, 259))                                  # an affinity mask that does not start at 0
    blocks = [shard.rank_cpus(r, 8, cpus) for r in range(8)]
    assert all(len(b) == 32 for b in blocks) and blocks[0][0] == 3 and blocks[7][-1] == 258
    assert sorted(c for b in blocks for c in b) == cpus
    assert shard.rank_cpus(1, 3, list(range(10))) == [3, 4, 5]  # 10 // 3 = 3 each, one cpu left unused
    assert shard.rank_cpus(0, 8, [0, 1]) == [0, 1]              # fewer cpus than ranks: no pinning
    assert shard.pin_rank(0, 1) is None                         # a single rank is left alone


def test_numa_rank_cpus_follow_the_gpus_nodes():
    """shard.numa_rank_cpus: a rank's cpus come from the NUMA node of ITS GPU, shared equally by the ranks on that node; unknown topology -> None
    (pin_rank then falls back to equal blocks)."""
    node_cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    mask = list(range(256))
    blocks = [shard.numa_rank_cpus(r, nodes, node_cpus, mask) for r in range(8)]
    assert all(len(b) == 32 for b in blocks)
    assert len(set().union(*blocks)) == 256
    assert all(set(blocks[r]) <= set(node_cpus[nodes[r]]) for r in range(8))
    assert blocks[0] == list(range(0, 32)) and blocks[4] == list(range(64, 96))
    assert shard.numa_rank_cpus(1, [0, 1], node_cpus, list(range(0, 100))) == list(range(64, 100))     # only the cpus the mask allows
    assert shard.numa_rank_cpus(0, [0, -1], node_cpus, mask) is None
    assert shard.numa_rank_cpus(0, None, node_cpus, mask) is None
    assert shard.numa_rank_cpus(0, [0, 0], {0: [5]}, mask) is None                                      # a share would be empty
    assert shard._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
