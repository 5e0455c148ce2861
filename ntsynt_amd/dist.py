"""Multi-GPU sharding of the hot path: one process per GPU, torch.distributed over RCCL/xGMI
(SURVEY.md 8(e)).  Genomes are partitioned over ranks; two exchange steps exist:

  1. common Bloom filter = bitwise AND of the per-genome filters.  RCCL has no
     bitwise reduction, so the all-reduce is: direct reduce-scatter (every rank sends chunk j of its
     filter to rank j over the i<->j xGMI link, all links busy at once), a local AND kernel over the
     received chunks, then an all-gather of the reduced chunks.
  2. all-gather(v) of the per-genome minimizer lists before the (replicated) graph stage.

Nothing here computes on the CPU: the AND operator is passed in by the caller (the HIP kernel via
nts_and_raw on GPU ranks; the gloo tests pass a tensor op to exercise the schedule)."""
import torch
import torch.distributed as dist


def genomes_of_rank(n_genomes, rank, world):
    "genome g -> rank g mod world"
    return [g for g in range(n_genomes) if g % world == rank]


def padded_len(nbytes, world):
    "buffer length that splits into `world` chunks of a multiple of 16 bytes"
    q = 16 * world
    return (int(nbytes) + q - 1) // q * q


def allreduce_and(buf, and_into, group=None):
    """In-place bitwise-AND all-reduce of the 1-D uint8 tensor `buf` (length = padded_len(...)).
    and_into(acc, other) must perform acc &= other on equal-length views of the tensors given."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return buf
    rank = dist.get_rank(group)
    n = buf.numel()
    assert n % (16 * world) == 0, "use padded_len() to size the buffer"
    chunk = n // world
    recv = torch.empty((world - 1) * chunk, dtype=buf.dtype, device=buf.device)
    ops, slot = [], {}
    for step in range(1, world):
        peer_to = (rank + step) % world
        peer_from = (rank - step) % world
        slot[peer_from] = len(slot)
        ops.append(dist.P2POp(dist.isend, buf[peer_to * chunk:(peer_to + 1) * chunk], peer_to, group))
        ops.append(dist.P2POp(dist.irecv, recv[slot[peer_from] * chunk:(slot[peer_from] + 1) * chunk], peer_from, group))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if buf.is_cuda:
        torch.cuda.synchronize(buf.device)
    mine = buf[rank * chunk:(rank + 1) * chunk]
    for s in range(world - 1):
        and_into(mine, recv[s * chunk:(s + 1) * chunk])
    gathered = torch.empty_like(buf)
    dist.all_gather_into_tensor(gathered, mine.contiguous(), group=group)
    buf.copy_(gathered)
    return buf


def allgather_lists(h1, rec, pos, genome_id, group=None):
    """All-gather(v) of minimizer lists.  Inputs are 1-D tensors of one local genome's list
    (h1 int64-viewed uint64, rec int32, pos int64) on the collective's device.  Returns a list of
    (genome_id, h1, rec, pos) for every rank's contribution, in rank order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return [(genome_id, h1, rec, pos)]
    dev = h1.device
    meta = torch.tensor([h1.numel(), genome_id], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0]) for m in metas]
    gids = [int(m[1]) for m in metas]
    cap = max(max(counts), 1)

    def gather(t, dtype):
        padded = torch.zeros(cap, dtype=dtype, device=dev)
        padded[:t.numel()] = t
        out = torch.empty(world * cap, dtype=dtype, device=dev)
        dist.all_gather_into_tensor(out, padded, group=group)
        return out

    gh, gr, gp = gather(h1, torch.int64), gather(rec, torch.int32), gather(pos, torch.int64)
    res = []
    for r in range(world):
        sl = slice(r * cap, r * cap + counts[r])
        res.append((gids[r], gh[sl], gr[sl], gp[sl]))
    return res
