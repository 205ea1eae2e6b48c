"""The multi-GPU path (row partition + gather to rank 0 + un-permute) on CPU with two processes over
gloo: same sharding.FrameGather the benchmark uses with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ltrace
import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _pixel_value(rows, width, channels):
    """A frame whose content encodes (global row, column, channel): any misplaced row shows."""
    r = rows.to(torch.int64)[:, None, None]
    x = torch.arange(width)[None, :, None]
    ch = torch.arange(channels)[None, None, :]
    return ((r * 131 + x * 7 + ch * 3) % 251).to(torch.uint8)


def _worker(rank, world, port, height, width, row_block, out_path, owner=None):
    for p in (ROOT, os.path.join(ROOT, "light-path-tracer_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        fg = sharding.FrameGather(height, width, 4, torch.uint8, "cpu", row_block, world, rank, owner=owner)
        rows = (sharding.global_row_index(height, row_block, world, rank) if owner is None
                else torch.from_numpy(ltrace.owned_rows(height, row_block, owner, rank)))
        assert fg.local_view().shape[0] == rows.numel()
        fg.local_view().copy_(_pixel_value(rows, width, 4))      # "render" this rank's rows
        for _ in range(2):                                        # buffers are reusable across frames
            full = fg.gather()
        if rank == 0:
            expect = _pixel_value(torch.arange(height), width, 4)
            assert torch.equal(full, expect)
            np.save(out_path, np.array([1]))
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("height,width,row_block", [(64, 40, 16), (37, 9, 8), (5, 3, 16)])
def test_two_rank_gather_reassembles_the_frame(tmp_path, height, width, row_block):
    world = 2
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(world, _free_port(), height, width, row_block, out), nprocs=world, join=True)
    assert os.path.exists(out)


def test_partition_covers_every_row_once():
    for h, rb, w in [(4096, 16, 8), (8192, 16, 8), (33, 16, 2), (100, 7, 3), (5, 16, 8)]:
        assert sharding.reference_partition_check(h, rb, w)
        assert sum(sharding.local_rows(h, rb, w, r) for r in range(w)) == h
    # the benchmark frame splits evenly at 1, 2, 4, 8 ranks
    for w in (1, 2, 4, 8):
        assert {ltrace.local_rows(4096, 16, w, r) for r in range(w)} == {4096 // w}


def test_two_rank_gather_with_a_block_owner_table(tmp_path):
    """The cost-weighted partition (lt_opts.block_owner / sharding.balance_blocks) through the same gather."""
    height, width, rb = 70, 12, 8
    owner = np.array([1, 1, 0, 1, 0, 0, 0, 1, 1], dtype=np.uint16)       # 9 blocks, the last one short
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, _free_port(), height, width, rb, out, owner), nprocs=2, join=True)
    assert os.path.exists(out)


def test_three_rank_gather_unpadded_with_unequal_and_empty_partitions(tmp_path):
    """Exact-size exchange: partitions of 48, 22 and 0 rows (a rank may own nothing) -- nothing is padded to the largest."""
    height, width, rb = 70, 12, 8
    owner = np.array([0, 0, 1, 0, 0, 1, 0, 0, 1], dtype=np.uint16)       # rank 2 owns no block; the last block is short
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(3, _free_port(), height, width, rb, out, owner), nprocs=3, join=True)
    assert os.path.exists(out)
    fg = sharding.FrameGather(height, width, 4, torch.uint8, "cpu", rb, 3, 1, owner=owner)
    assert fg.rows == [48, 22, 0] and tuple(fg.local.shape) == (22, width, 4)   # a peer allocates its own rows only


def test_padded_diagnostic_exchange_gives_the_same_frame(tmp_path, monkeypatch):
    monkeypatch.setenv("LT_FRAME_GATHER", "padded")
    owner = np.array([1, 1, 0, 1, 0, 0, 0, 1, 1], dtype=np.uint16)
    out = str(tmp_path / "ok.npy")
    mp.spawn(_worker, args=(2, _free_port(), 70, 12, 8, out, owner), nprocs=2, join=True)
    assert os.path.exists(out)


def test_a_table_that_is_not_a_partition_is_refused():
    with pytest.raises(ValueError):
        sharding.FrameGather(64, 8, 4, torch.uint8, "cpu", 16, 2, 0, owner=np.array([0, 0, 0], dtype=np.uint16))   # 4 blocks, 3 entries


def test_balance_blocks_properties():
    rng = np.random.default_rng(0)
    nb = 256
    cost = rng.integers(9_000_000, 11_000_000, nb).astype(float)
    chain = rng.integers(200, 400, nb).astype(float)
    chain[[5, 77, 130, 200]] = [7484, 7329, 7111, 6822]
    for world in (1, 2, 4, 8):
        o = sharding.balance_blocks(cost, chain, world, chain_cost=137500.0)
        assert o.dtype == np.uint16 and o.shape == (nb,) and o.max() < world
        assert np.array_equal(o, sharding.balance_blocks(cost, chain, world, chain_cost=137500.0))   # deterministic
        if world == 8:
            # the four long chains land on four different ranks, and those ranks get less of the bulk
            owners = o[[5, 77, 130, 200]]
            assert len(set(owners.tolist())) == 4
            bulk = np.array([cost[o == r].sum() for r in range(world)])
            assert bulk[owners].max() < bulk[[r for r in range(world) if r not in owners]].min()
