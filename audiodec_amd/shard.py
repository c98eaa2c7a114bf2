"""Multi-GPU: independent streams are the shard.

One process per GPU (torch.distributed, backend "nccl" = RCCL on ROCm).  Stream k of a job lives
on rank ``k // streams_per_rank``; every rank holds a full weight replica and its own state
arena, so the steady state needs NO collective.  RCCL is used only for
  * ``broadcast_state_dict``: rank 0 loads (or synthesises) a checkpoint, everyone receives the
    same bits (one flat fp32 broadcast per state dict),
  * ``gather_codes`` / ``gather_rows``: optional egress of emitted indices / waveforms to rank 0,
  * the timing barrier / max-reduction in bench.py.
The same functions run on CPU tensors over gloo (tests/test_shard_gloo.py).
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def stream_range(n_streams, rank=None, world_size=None):
    """Contiguous block of global stream ids owned by `rank` (sizes differ by at most one)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    base, extra = divmod(n_streams, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def owner_of(stream, n_streams, world_size):
    for r in range(world_size):
        lo, hi = stream_range(n_streams, r, world_size)
        if lo <= stream < hi:
            return r
    raise IndexError(stream)


def broadcast_state_dict(sd, src=0, device=None):
    """Make every rank hold rank `src`'s tensors.  Non-src ranks pass a dict with the same keys/shapes
    (e.g. freshly constructed) or None together with `template` semantics handled by the caller."""
    rank, w = world()
    if w == 1:
        return sd
    keys = sorted(sd.keys())
    dev = device if device is not None else "cpu"
    flat = torch.cat([sd[k].detach().reshape(-1).to(torch.float32) for k in keys]).to(dev)
    dist.broadcast(flat, src=src)
    flat = flat.cpu()
    out, off = {}, 0
    for k in keys:
        n = sd[k].numel()
        out[k] = flat[off:off + n].reshape(sd[k].shape).to(sd[k].dtype)
        off += n
    return out


def gather_rows(local, dst=0):
    """Concatenate per-rank tensors along dim 0 on `dst` (ranks may own different stream counts)."""
    rank, w = world()
    if w == 1:
        return local
    sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(w)]
    dist.all_gather(sizes, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device))
    mx = int(max(int(s) for s in sizes))
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(bufs, pad)
    if rank != dst:
        return None
    return torch.cat([b[:int(s)] for b, s in zip(bufs, sizes)], 0)


def gather_codes(idx, dst=0):
    """idx (n_q, B_local, T) -> (n_q, B_total, T) on `dst`."""
    g = gather_rows(idx.transpose(0, 1).contiguous(), dst)
    return None if g is None else g.transpose(0, 1).contiguous()


def rank_cpus(local_rank, local_world, cpus=None):
    """The host cpus rank `local_rank` of `local_world` ranks on this node should run on: a contiguous block of the cpus this process
    may use (its affinity mask), equal sizes, the remainder left unused.  One process per GPU feeds ~40 launches per step from Python;
    eight unpinned ranks on one host migrate between cores and share them with each other's runtime threads ("the risk is
    host-side feeding").  Pure function of its arguments when `cpus` is given (tests)."""
    import os
    if cpus is None:
        cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    cpus = list(cpus)
    per = len(cpus) // max(1, local_world)
    if per < 1:
        return cpus                          # fewer cpus than ranks: no pinning
    return cpus[local_rank * per:(local_rank + 1) * per]


def numa_rank_cpus(local_rank, gpu_nodes, node_cpus, cpus):
    """NUMA-aware variant of rank_cpus, pure (tests): gpu_nodes[r] = NUMA node of the GPU of local rank r (-1 / None: unknown),
    node_cpus[n] = cpus of node n, cpus = the affinity mask.  Rank r gets an equal share of ITS GPU's node's cpus (those in the mask),
    split among the local ranks whose GPUs sit on the same node, in rank order -- the launches of a rank then go out from cores next to
    its GPU's PCIe root.  Returns None when any rank's node is unknown or a share would be empty: the caller falls back to rank_cpus()."""
    if not gpu_nodes or any(n is None or n < 0 for n in gpu_nodes) or not 0 <= local_rank < len(gpu_nodes):
        return None
    allowed = set(cpus)
    shares = {}
    for node in set(gpu_nodes):
        mine = [c for c in sorted(node_cpus.get(node, [])) if c in allowed]
        ranks = [r for r, n in enumerate(gpu_nodes) if n == node]
        per = len(mine) // len(ranks)
        if per < 1:
            return None
        for k, r in enumerate(ranks):
            shares[r] = mine[k * per:(k + 1) * per]
    return shares[local_rank]


def _parse_cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(n_gpus):
    """(NUMA node of each of the first n_gpus HIP devices, {node: cpus}) from sysfs -- the PCI address torch reports for a device ->
    /sys/bus/pci/devices/<addr>/numa_node, /sys/devices/system/node/node<N>/cpulist; (None, {}) wherever any of that is missing."""
    import os
    try:
        nodes = []
        for i in range(n_gpus):
            pr = torch.cuda.get_device_properties(i)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            with open(f"/sys/bus/pci/devices/{addr}/numa_node") as f:
                nodes.append(int(f.read().strip()))
        node_cpus = {}
        for n in set(nodes):
            if n < 0:
                return None, {}
            with open(f"/sys/devices/system/node/node{n}/cpulist") as f:
                node_cpus[n] = _parse_cpulist(f.read())
        return nodes, node_cpus
    except Exception:
        return None, {}


def pin_rank(local_rank, local_world, one_gpu=False):
    """Restrict this process to its share of the host's cpus (and tell the intra-op pools); returns the cpu list, or None where the
    platform has no affinity call or the mask could not be set.  The share: the cpus of the NUMA node of the rank's GPU
    (/sys/bus/pci/devices/<gpu>/numa_node), divided among the ranks whose GPUs share that node (numa_rank_cpus) -- when the topology
    cannot be read, or all ranks share one GPU (one_gpu: the rehearsal on a 1-GPU box), equal contiguous blocks of the affinity mask
    (rank_cpus)."""
    import os
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = None
    if not one_gpu and torch.cuda.is_available() and torch.cuda.device_count() >= local_world:
        nodes, node_cpus = gpu_numa_nodes(local_world)
        if nodes is not None:
            cpus = numa_rank_cpus(local_rank, nodes, node_cpus, sorted(os.sched_getaffinity(0)))
    if not cpus:
        cpus = rank_cpus(local_rank, local_world)
    try:
        os.sched_setaffinity(0, cpus)
    except OSError:
        return None
    torch.set_num_threads(max(1, min(len(cpus), 8)))
    return cpus


def max_over_ranks(value, device):
    rank, w = world()
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if w > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, device):
    """One float per rank, on every rank (bench.py: each rank's own frames/s beside the whole-job figure)."""
    rank, w = world()
    if w == 1:
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    bufs = [torch.empty_like(t) for _ in range(w)]
    dist.all_gather(bufs, t)
    return [float(b.item()) for b in bufs]
