"""Multi-GPU plumbing: one process per GPU, the batch axis sharded across ranks.

The reference has no distributed layer: its only scaling axis is the batch, spread over threads
(src/libxsmm_gemm.c:1321-1324 gives thread `tid` the slice [tid*tasksize, (tid+1)*tasksize)). The same contiguous block
partition is used across GPUs here. Items are independent, so compute needs no data-path collective; two optional
epilogues exist:
  * all-gather of per-shard C blocks when every rank needs the whole result (BASELINE config 4),
  * sum-reduction of partial C blocks when products of one C block were split over ranks (CP2K / blocked_gemm, config 5).
Backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" on CPUs (tests). torch.distributed is plumbing only.
"""
import os


def shard_range(n, rank, world):
    """Contiguous block partition of n items: the reference's tasksize rule with ntasks = world (libxsmm_gemm.c:1321-1324)."""
    tasksize = (n + world - 1) // world
    begin = min(rank * tasksize, n)
    return begin, min(begin + tasksize, n)


def shard_by_c_owner(c_ids, world):
    """Reduction workloads: assign every product to the rank that owns its C block (owner = block partition of the sorted
    distinct C ids), so that no exchange is needed. Returns a list of index lists, one per rank (batch order kept)."""
    distinct = sorted(set(int(c) for c in c_ids))
    owner = {}
    for r in range(world):
        b, e = shard_range(len(distinct), r, world)
        for c in distinct[b:e]:
            owner[c] = r
    out = [[] for _ in range(world)]
    for i, c in enumerate(c_ids):
        out[owner[int(c)]].append(i)
    return out


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK/WORLD_SIZE/MASTER_*). Returns (rank, world, dist|None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world <= 1:
        return 0, 1, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend=backend, **kw)
    return rank, world, dist


def allgather_shards(local, n_total, dist):
    """Every rank contributes the C blocks of its shard (flat tensor, shard sizes follow shard_range) and receives the
    concatenation. One all_gather_into_tensor over equal, padded shards: on xGMI every GPU pushes its shard to the 7
    peers concurrently, so a few large messages beat many small ones."""
    import torch
    if dist is None:
        return local
    world = dist.get_world_size()
    per_item = local.numel() // max(1, (shard_range(n_total, dist.get_rank(), world)[1] - shard_range(n_total, dist.get_rank(), world)[0]))
    tasksize = (n_total + world - 1) // world
    padded = torch.zeros(tasksize * per_item, dtype=local.dtype, device=local.device)
    padded[:local.numel()] = local
    out = torch.empty(world * tasksize * per_item, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return out[:n_total * per_item]


def gather_chunks_overlapped(nchunks, compute_chunk, local_chunk, gathered_chunk, dist):
    """BASELINE config 4 epilogue, overlapped: a rank's shard of C is produced chunk by chunk; while chunk i + 1 is being
    computed, chunk i is all-gathered (`gathered_chunk(i)`: world x chunk elements, `local_chunk(i)`: this rank's chunk; equal
    chunk sizes on all ranks). With the nccl (= RCCL) backend the collective runs on RCCL's own stream behind an event of the
    compute stream, so the xGMI transfer of one chunk hides behind the kernels of the next; per-link bound (7 x ~153 GB/s per
    GPU), hence few large chunks. Returns when every chunk has arrived (the compute stream waits for the collectives)."""
    works = []
    for i in range(nchunks):
        compute_chunk(i)
        if dist is not None:
            works.append(dist.all_gather_into_tensor(gathered_chunk(i), local_chunk(i), async_op=True))
    for w in works:
        w.wait()


def owned_c_blocks(n_blocks, rank, world):
    """BASELINE config 5, preferred partition: rank r owns the contiguous share shard_range(n_blocks, r, world) of the C blocks
    of every shape group and is handed exactly the products that update them -- no exchange at all."""
    return shard_range(n_blocks, rank, world)


def reduce_partial_c(partial, dist):
    """Sum partial C blocks over ranks (one fused all-reduce over the concatenated block array)."""
    if dist is not None:
        dist.all_reduce(partial, op=dist.ReduceOp.SUM)
    return partial


def max_over_ranks(value, dist, device="cpu"):
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
