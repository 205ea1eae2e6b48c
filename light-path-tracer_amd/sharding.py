"""Row sharding of a frame across the GPUs of one node, one process per GPU.

The frame tiles trivially: pixels are independent, so ranks share no data while rendering and the
only exchange is the gather of the finished framebuffer to rank 0 (RCCL over xGMI when the process
group's backend is "nccl"; the same code runs on "gloo" for the CPU tests).

Partition: block-cyclic rows -- block b of `row_block` rows belongs to rank b % world.  Row cost is
far from uniform (shadow rows are cheap, critical-curve rows expensive), so contiguous slabs would
leave most GPUs idle; small cyclic blocks give every rank the same mix.
"""
import numpy as np
import torch
import torch.distributed as dist

import ltrace


def local_rows(height, row_block, world, rank):
    """Number of rows rank `rank` owns."""
    return ltrace.local_rows(height, row_block, world, rank)


def global_row_index(height, row_block, world, rank):
    """int64 tensor: global row of every local row of `rank` (ascending)."""
    return torch.from_numpy(ltrace.global_rows(height, row_block, world, rank))


def balance_blocks(block_cost, block_chain, world, chain_cost=1.0, share=0.65):
    """Cost-weighted assignment of row blocks to ranks (uint16 owner table for lt_opts.block_owner).

    A rank's frame time is bounded below by the longest serial chain it owns -- one ray of thousands of RK4 steps
    that no amount of hardware shortens -- plus what that ray loses while the rank's bulk shares the
    chip with it.  Block-cyclic assignment gives every rank the same bulk, so the rank that happens to own the
    longest ray finishes last.  Given the previous frame's per-block totals (`block_cost`, e.g. sum of steps) and
    longest ray (`block_chain`, max of steps), blocks are dealt out greedily, most expensive first, each to the rank
    whose modelled time  max(bulk, chain * chain_cost + share * bulk)  grows least; `chain_cost` converts a chain
    step into the units of block_cost (time of one lone step / time of one bulk step-lane).
    Deterministic: every rank computes the same table from the same (all-gathered) inputs."""
    cost = np.asarray(block_cost, dtype=np.float64)
    chain = np.asarray(block_chain, dtype=np.float64) * chain_cost
    nb = cost.size
    owner = np.zeros(nb, dtype=np.uint16)
    bulk = np.zeros(world)
    longest = np.zeros(world)

    def t(b, c):
        return np.maximum(b, c + share * b)

    order = np.lexsort((np.arange(nb), -cost, -chain))      # longest chains first, then by cost; ties by index
    for b in order:
        new_t = t(bulk + cost[b], np.maximum(longest, chain[b]))
        worst_if = np.array([max(new_t[r], np.delete(t(bulk, longest), r).max(initial=0.0)) for r in range(world)])
        r = int(np.lexsort((np.arange(world), bulk, worst_if))[0])
        owner[b] = r
        bulk[r] += cost[b]
        longest[r] = max(longest[r], chain[b])
    return owner


class FrameGather:
    """Gathers per-rank row partitions (R_rank, W, C) of one dtype to the full (H, W, C) frame on rank 0.

    Buffers are allocated once; `gather(local)` is collective.  Partitions are padded to the largest
    partition so a plain dist.gather (ncclGather-style send/recv into rank 0) suffices.
    `owner`: optional row-block -> rank table (balance_blocks); default block-cyclic."""

    def __init__(self, height, width, channels, dtype, device, row_block, world=None, rank=None, owner=None):
        self.world = dist.get_world_size() if world is None else world
        self.rank = dist.get_rank() if rank is None else rank
        self.h, self.w, self.c = height, width, channels
        self.row_block = row_block
        self.device = torch.device(device)
        self.owner = None if owner is None else np.ascontiguousarray(owner, dtype=np.uint16)
        if self.owner is not None:
            self._rows_of = [torch.from_numpy(ltrace.owned_rows(height, row_block, self.owner, r)) for r in range(self.world)]
            self.rows = [int(x.numel()) for x in self._rows_of]
        else:
            self._rows_of = None
            self.rows = [local_rows(height, row_block, self.world, r) for r in range(self.world)]
        self.rows_max = max(self.rows)
        self.local = torch.empty((self.rows_max, width, channels), dtype=dtype, device=self.device)
        self.full = None
        self.parts = None
        if self.rank == 0:
            self.full = torch.empty((height, width, channels), dtype=dtype, device=self.device)
            if self.world > 1:
                self.parts = [torch.empty_like(self.local) for _ in range(self.world)]
        self._index = None
        if self._rows_of is not None and self.rank == 0:
            self._index = [x.to(self.device) for x in self._rows_of]

    def local_view(self):
        """(R_rank, W, C) view this rank renders into."""
        return self.local[: self.rows[self.rank]]

    def gather(self, stream_ptr=0):
        """Collective.  Rank 0 returns the assembled (H, W, C) frame, other ranks None."""
        if self.world == 1:
            # one partition = the whole frame, already in row order
            return self.local[: self.h]
        if self.device.type == "cuda" and dist.get_backend() == "gloo":
            # rehearsals only (bench.py --backend gloo on a box with fewer GPUs than ranks): gloo gathers host tensors
            host = [torch.empty_like(self.local, device="cpu") for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(self.local.cpu(), host, dst=0)
            if self.rank == 0:
                for p_, h_ in zip(self.parts, host):
                    p_.copy_(h_)
        else:
            dist.gather(self.local, self.parts, dst=0)
        if self.rank != 0:
            return None
        elem = self.local.element_size() * self.c
        for r in range(self.world):
            if self.rows[r] == 0:
                continue
            if self.device.type == "cuda" and self.owner is None:
                ltrace.scatter_rows_dev(self.parts[r].data_ptr(), self.full.data_ptr(), self.h, self.w, elem,
                                        self.row_block, self.world, r, stream_ptr)
            else:
                if self._index is None:
                    self._index = [global_row_index(self.h, self.row_block, self.world, q).to(self.device)
                                   for q in range(self.world)]
                self.full.index_copy_(0, self._index[r], self.parts[r][: self.rows[r]])   # table mode, or CPU
        return self.full


def reference_partition_check(height, row_block, world):
    """All ranks' rows form a partition of range(height) (host-only helper for tests)."""
    seen = np.zeros(height, dtype=np.int64)
    for r in range(world):
        seen[ltrace.global_rows(height, row_block, world, r)] += 1
    return bool(np.all(seen == 1))
