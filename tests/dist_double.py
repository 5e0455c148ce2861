"""TEST INFRASTRUCTURE (moved out of the package in round 4: the product's exchanges are nts_bf_allreduce_and / _groups and
nts_mx_allgather in csrc/nts_comm.inc, and nothing else): a torch.distributed statement of the same two exchange schedules, with
which the CPU test doubles of tests/test_dist_*.py exercise ntsynt_amd.pipeline's multi-rank orchestration over gloo.

Multi-GPU sharding of the hot path: genomes are partitioned over ranks; two exchange steps exist:

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


class PackedListGather:
    """Exchange 2 for a steady stream of steps (bench.py): every rank's minimizer lists travel in ONE all-gather per
    step -- fixed-capacity slots in a single buffer, counts in a header, no size exchange -- and the collective of
    step i runs behind the sketch kernels of step i+1 (async on the collective's own stream, two buffer sets).

    Buffer (int64 words): [2 * n_lists header: count, genome id] then per list a slot of h1[cap] | pos[cap] |
    rec[cap] (int32, packed two per word)."""

    def __init__(self, n_lists, cap, device, comm_device=None, group=None):
        self.n_lists, self.cap, self.group = int(n_lists), int(cap), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.slot_words = 2 * self.cap + (self.cap + 1) // 2
        self.words = 2 * self.n_lists + self.n_lists * self.slot_words
        self.device = torch.device(device)
        self.comm_device = torch.device(comm_device or device)
        self.send = [torch.zeros(self.words, dtype=torch.int64, device=self.device) for _ in range(2)]
        self.stage = [None, None]
        if self.comm_device != self.device:                # verification mode: collectives on host copies
            self.stage = [torch.zeros(self.words, dtype=torch.int64, device=self.comm_device) for _ in range(2)]
        self.recv = [torch.empty(self.world * self.words, dtype=torch.int64, device=self.comm_device) for _ in range(2)]
        self.pending = [None, None]
        self.header = torch.zeros(2 * self.n_lists, dtype=torch.int64)
        self.turn = 0

    def slot_ptrs(self, i):
        "device addresses (h1, rec, pos) of list i's slot in the buffer being filled"
        base = self.send[self.turn].data_ptr() + (2 * self.n_lists + i * self.slot_words) * 8
        return base, base + 2 * self.cap * 8, base + self.cap * 8

    def begin(self):
        "before filling the next buffer: its previous collective (two steps back) must be over"
        self._wait(self.turn)

    def set_count(self, i, count, genome_id):
        if count > self.cap:
            raise RuntimeError(f"minimizer list of {count} entries exceeds the exchange slot ({self.cap})")
        self.header[2 * i], self.header[2 * i + 1] = int(count), int(genome_id)

    def post(self):
        "header in, collective out (asynchronous); returns at once"
        t = self.turn
        self.send[t][:2 * self.n_lists].copy_(self.header)
        src = self.send[t]
        if self.stage[t] is not None:
            self.stage[t].copy_(src)
            src = self.stage[t]
        if self.world > 1:
            self.pending[t] = dist.all_gather_into_tensor(self.recv[t], src, group=self.group, async_op=True)
        self.turn ^= 1

    def _wait(self, t):
        if self.pending[t] is not None:
            self.pending[t].wait()
            self.pending[t] = None
            if self.comm_device.type == "cuda":
                torch.cuda.current_stream(self.comm_device).synchronize()

    def drain(self):
        self._wait(0)
        self._wait(1)

    def lists_of(self, t):
        "decode buffer set t after drain(): [(genome id, h1, rec, pos)] of every rank, as views"
        out = []
        buf = self.recv[t] if self.world > 1 else self.send[t]
        for r in range(self.world):
            b = buf[r * self.words:(r + 1) * self.words]
            for i in range(self.n_lists):
                n, gid = int(b[2 * i]), int(b[2 * i + 1])
                s = b[2 * self.n_lists + i * self.slot_words:][:self.slot_words]
                rec = s[2 * self.cap:].view(torch.int32)[:n]
                out.append((gid, s[:n], rec, s[self.cap:self.cap + n]))
        return out
