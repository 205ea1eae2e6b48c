"""Row sharding of a frame across the GPUs of one node, one process per GPU.

The frame tiles trivially: pixels are independent, so ranks share no data while rendering and the
only exchange is the finished framebuffer going to rank 0 (RCCL send/recv over xGMI when the process
group's backend is "nccl"; the same code runs on "gloo" for the CPU tests).

Partition: block-cyclic rows -- block b of `row_block` rows belongs to rank b % world.  Row cost is
far from uniform (shadow rows are cheap, critical-curve rows expensive), so contiguous slabs would
leave most GPUs idle; small cyclic blocks give every rank the same mix.
"""
import os

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
    """Brings per-rank row partitions (R_rank, W, C) of one dtype together as the full (H, W, C) frame on rank 0.

    Every rank sends exactly the rows it owns -- one grouped point-to-point exchange (`dist.batch_isend_irecv`: on the
    "nccl" backend a single ncclGroupStart/End of ncclSend / ncclRecv over xGMI, the seven peers of an 8-GPU node on
    seven links into rank 0 at once) -- into a rank-major staging frame on rank 0: rank r's rows at
    [offset_r, offset_r + rows_r).  Rank 0 renders straight into its own piece of it.  One scatter launch
    (lt_scatter_rows_indexed_dev: staging row i -> frame row index[i]) then un-permutes the whole frame, whatever the
    assignment of row blocks to ranks.  Nothing is padded: with a cost-weighted table the partitions differ by up to
    3.6x (224 ... 816 rows of 4096) and a padded `dist.gather` would move 1.6x the frame.
    Buffers are allocated once; `gather()` is collective.
    `owner`: optional row-block -> rank table (balance_blocks); default block-cyclic.
    LT_FRAME_GATHER=padded in the environment selects the old exchange (dist.gather of partitions padded to the
    largest) -- a diagnostic switch for a transport that mishandles grouped send/recv, not a second product path."""

    def __init__(self, height, width, channels, dtype, device, row_block, world=None, rank=None, owner=None):
        self.world = dist.get_world_size() if world is None else world
        self.rank = dist.get_rank() if rank is None else rank
        self.h, self.w, self.c = height, width, channels
        self.row_block = row_block
        self.device = torch.device(device)
        self.owner = None if owner is None else np.ascontiguousarray(owner, dtype=np.uint16)
        if self.owner is not None:
            rows_of = [ltrace.owned_rows(height, row_block, self.owner, r) for r in range(self.world)]
        else:
            rows_of = [ltrace.global_rows(height, row_block, self.world, r) for r in range(self.world)]
        self.rows = [int(x.size) for x in rows_of]
        if sum(self.rows) != height or not np.array_equal(np.sort(np.concatenate(rows_of)), np.arange(height)):
            raise ValueError("the row partition does not cover every row of the frame exactly once")
        self.offsets = [int(x) for x in np.concatenate([[0], np.cumsum(self.rows)])]
        self.rows_max = max(self.rows)
        self.padded = os.environ.get("LT_FRAME_GATHER", "") == "padded" and self.world > 1
        self.full = self.staging = self.index = self.parts = None
        if self.world == 1:
            self.local = torch.empty((self.rows[0], width, channels), dtype=dtype, device=self.device)
        elif self.padded:
            self.local = torch.empty((self.rows_max, width, channels), dtype=dtype, device=self.device)
            if self.rank == 0:
                self.parts = [torch.empty_like(self.local) for _ in range(self.world)]
        elif self.rank == 0:
            self.staging = torch.empty((height, width, channels), dtype=dtype, device=self.device)
            self.local = self.staging[: self.rows[0]]
        else:
            self.local = torch.empty((self.rows[self.rank], width, channels), dtype=dtype, device=self.device)
        if self.rank == 0 and self.world > 1:
            self.full = torch.empty((height, width, channels), dtype=dtype, device=self.device)
            self.index = torch.from_numpy(np.concatenate(rows_of).astype(np.int64)).to(self.device)
            self._index_of = [torch.from_numpy(x.astype(np.int64)).to(self.device) for x in rows_of] if self.padded else None

    def local_view(self):
        """(R_rank, W, C) view this rank renders into."""
        return self.local[: self.rows[self.rank]]

    def _exchange(self):
        staged = self.device.type == "cuda" and dist.get_backend() == "gloo"   # rehearsals only: gloo moves host tensors
        if self.padded:
            if staged:
                host = [torch.empty_like(self.local, device="cpu") for _ in range(self.world)] if self.rank == 0 else None
                dist.gather(self.local.cpu(), host, dst=0)
                if self.rank == 0:
                    for p_, h_ in zip(self.parts, host):
                        p_.copy_(h_)
            else:
                dist.gather(self.local, self.parts, dst=0)
            return
        if self.rank == 0:
            pieces = [(r, self.staging[self.offsets[r]: self.offsets[r + 1]]) for r in range(1, self.world) if self.rows[r]]
            bufs = [(r, torch.empty(v.shape, dtype=v.dtype, device="cpu") if staged else v) for r, v in pieces]
            ops = [dist.P2POp(dist.irecv, b, r) for r, b in bufs]
            if ops:
                for wk in dist.batch_isend_irecv(ops):
                    wk.wait()
            if staged:
                for (_, v), (_, b) in zip(pieces, bufs):
                    v.copy_(b)
        elif self.rows[self.rank]:
            src = self.local_view()
            for wk in dist.batch_isend_irecv([dist.P2POp(dist.isend, src.cpu() if staged else src, 0)]):
                wk.wait()

    def gather(self, stream_ptr=0):
        """Collective.  Rank 0 returns the assembled (H, W, C) frame, other ranks None.
        With a CUDA device the caller's current torch stream must be the stream behind `stream_ptr` (bench.py wraps the
        call in torch.cuda.stream): the exchange is ordered by torch on the current stream, the scatter by the pointer."""
        if self.world == 1:
            return self.local[: self.h]          # one partition = the whole frame, already in row order
        self._exchange()
        if self.rank != 0:
            return None
        row_bytes = self.w * self.c * self.full.element_size()
        if self.padded:                          # diagnostic exchange: per-partition un-permute on the current stream
            for r in range(self.world):
                if self.rows[r]:
                    self.full.index_copy_(0, self._index_of[r], self.parts[r][: self.rows[r]])
        elif self.device.type == "cuda":
            ltrace.scatter_rows_indexed_dev(self.staging.data_ptr(), self.full.data_ptr(), self.index.data_ptr(), self.h, self.h,
                                            row_bytes, stream_ptr)
        else:
            self.full.index_copy_(0, self.index, self.staging)
        return self.full


def reference_partition_check(height, row_block, world):
    """All ranks' rows form a partition of range(height) (host-only helper for tests)."""
    seen = np.zeros(height, dtype=np.int64)
    for r in range(world):
        seen[ltrace.global_rows(height, row_block, world, r)] += 1
    return bool(np.all(seen == 1))
