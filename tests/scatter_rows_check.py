"""Run by tests/test_gpu_multi.py in its own process: the two row-scatter entry points of the C-ABI against numpy.
Row sizes that are and are not multiples of 16, bases that are and are not 16-byte aligned (the 16-byte-per-lane path must
not be taken then), a permutation with an out-of-range entry (skipped, not a fault)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "light-path-tracer_amd"))
import ltrace  # noqa: E402


def check(row_bytes, offset):
    rng = np.random.default_rng(5)
    H, rb, parts = 77, 8, 3
    dev = torch.device("cuda:0")
    frame = rng.integers(0, 256, size=(H, row_bytes), dtype=np.uint8)
    # indexed: rows in a shuffled order, one entry pointing outside the frame
    perm = rng.permutation(H).astype(np.int64)
    src = frame[perm]
    idx = perm.copy()
    idx[5] = H + 3
    pad_s = torch.zeros(src.size + 64, dtype=torch.uint8, device=dev)
    pad_d = torch.full((H * row_bytes + 64,), 7, dtype=torch.uint8, device=dev)
    pad_s[offset: offset + src.size] = torch.from_numpy(src.reshape(-1)).to(dev)
    d_idx = torch.from_numpy(idx).to(dev)
    torch.cuda.synchronize()
    ltrace.scatter_rows_indexed_dev(pad_s.data_ptr() + offset, pad_d.data_ptr() + offset, d_idx.data_ptr(), H, H, row_bytes)
    torch.cuda.synchronize()
    got = pad_d.cpu().numpy()
    want = frame.copy()
    want[perm[5]] = 7                                   # the skipped row keeps the fill value
    assert np.array_equal(got[offset: offset + H * row_bytes].reshape(H, row_bytes), want), ("indexed", row_bytes, offset)
    assert np.all(got[:offset] == 7) and np.all(got[offset + H * row_bytes:] == 7), ("indexed: wrote outside", row_bytes, offset)
    # block-cyclic: partition p's rows back into place (width x elem_bytes = row_bytes)
    pad_d.fill_(9)
    keep = []
    for p in range(parts):
        rows = ltrace.global_rows(H, rb, parts, p)
        part = torch.zeros(rows.size * row_bytes + 64, dtype=torch.uint8, device=dev)
        part[offset: offset + rows.size * row_bytes] = torch.from_numpy(frame[rows].reshape(-1)).to(dev)
        keep.append(part)
        torch.cuda.synchronize()
        ltrace.scatter_rows_dev(part.data_ptr() + offset, pad_d.data_ptr() + offset, H, row_bytes, 1, rb, parts, p)
    torch.cuda.synchronize()
    got = pad_d.cpu().numpy()
    assert np.array_equal(got[offset: offset + H * row_bytes].reshape(H, row_bytes), frame), ("cyclic", row_bytes, offset)
    assert np.all(got[:offset] == 9) and np.all(got[offset + H * row_bytes:] == 9), ("cyclic: wrote outside", row_bytes, offset)


if __name__ == "__main__":
    for rb_, off in ((256, 0), (4096 * 4, 0), (13, 0), (48, 1), (40, 8)):
        check(rb_, off)
    print("scatter_rows_check ok")
