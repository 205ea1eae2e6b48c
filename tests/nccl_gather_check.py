"""One rank of the multi-GPU parity check (started by tests/test_gpu_multi.py, one process per GPU): every rank renders its
block-cyclic row partition, the RGBA8 framebuffer is gathered to rank 0 over RCCL (sharding.FrameGather, backend nccl)
and un-permuted there; rank 0 also renders the whole frame alone and the two must agree byte for byte."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "light-path-tracer_amd")):
    sys.path.insert(0, p)

import numpy as np          # noqa: E402
import torch                # noqa: E402  (before ltrace: see tests/hipmini.py)
import torch.distributed as dist  # noqa: E402

import ltrace               # noqa: E402
import sharding             # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # LT_CHECK_DEVICES="0,0" + LT_CHECK_BACKEND=gloo: the same check with the ranks sharing devices and the gather staged
    # through the host (RCCL refuses two ranks on one device) -- what a 1-GPU box can run of the N-rank path
    dev_map = os.environ.get("LT_CHECK_DEVICES")
    index = int(dev_map.split(",")[rank]) if dev_map else rank
    torch.cuda.set_device(index)
    dev = torch.device("cuda", index)
    if os.environ.get("LT_CHECK_BACKEND", "nccl") == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(os.environ["LT_CHECK_BACKEND"])
    W, H, rb = 1000, 777, 16                      # ragged: partitions of unequal size, last block short
    fov_v = np.radians(40.0)
    cam = ltrace.Camera(W, H, 2 * np.arctan(np.tan(fov_v / 2) * W / H), fov_v, 0.02, -0.03, 50.0, np.pi / 2)
    met = ltrace.Metric(1, 0, 1.0, 0.9)
    stream = torch.cuda.current_stream(dev)
    whole = None
    if rank == 0:
        whole = torch.empty((H, W, 4), dtype=torch.uint8, device=dev)
        o1 = ltrace.default_opts(precision=32)
        o1.stream = stream.cuda_stream
        ltrace.render_dev(cam, met, o1, d_rgba=whole.data_ptr())
        torch.cuda.synchronize(dev)
    ok = True
    # block-cyclic, then a row-block -> rank table with partitions of very different sizes (what --balance cost produces)
    nb = (H + rb - 1) // rb
    table = (np.random.default_rng(5).integers(0, world, nb) * (np.arange(nb) % 3 != 0)).astype(np.uint16)
    for owner in (None, table):
        fg = sharding.FrameGather(H, W, 4, torch.uint8, dev, rb, world, rank, owner=owner)
        o = ltrace.default_opts(precision=32, n_parts=world, part=rank, row_block=rb, block_owner=owner)
        o.stream = stream.cuda_stream
        for _ in range(2):                            # buffers are reused frame after frame
            ltrace.render_dev(cam, met, o, d_rgba=fg.local_view().data_ptr())
            full = fg.gather(stream.cuda_stream)
        torch.cuda.synchronize(dev)
        if rank == 0:
            same = bool(torch.equal(full, whole))
            ok = ok and same
            print(f"{dist.get_backend()} exchange over {world} ranks, {'block-cyclic' if owner is None else 'table ' + str(fg.rows)}: "
                  f"frame {'identical' if same else 'DIFFERS'}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
