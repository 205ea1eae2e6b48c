#!/usr/bin/env python3
"""Does RCCL accept two ranks on ONE device (to rehearse the N-rank bench path on a 1-GPU box)?  Parent starts two children."""
import os, subprocess, sys
if "RANK" not in os.environ:
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rcs = []
    for p in procs:
        try:
            rcs.append(p.wait(timeout=120))
        except subprocess.TimeoutExpired:
            p.kill(); rcs.append(124)
    print("exit codes", rcs)
    sys.exit(max(rcs))
import torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
x = torch.tensor([float(dist.get_rank() + 1)], device=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
print("rank", dist.get_rank(), "all_reduce ->", x.item(), flush=True)
dist.destroy_process_group()
