import os, sys, time
sys.path.insert(0, os.getcwd())
n = int(sys.argv[1])
os.environ["OMP_NUM_THREADS"] = str(n)
import numpy as np
from oracle import oracle
fov = np.radians(40.0)
oracle.lookup("kerr", 1.0, 0.9, 50.0, 64, 64, fov, fov, integrator="rk4", perf_build=True)
t0 = time.perf_counter(); oracle.lookup("kerr", 1.0, 0.9, 50.0, 1024, 1024, fov, fov, integrator="rk4", perf_build=True); dt = time.perf_counter() - t0
print(f"threads {n}: {1024*1024/dt/1e6:.3f} Mrays/s ({dt:.2f} s)", flush=True)
