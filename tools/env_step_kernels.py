"""Which kernels a whole env.step() at the headline shape launches (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
bench.PLAN_WORKERS = 0
dev = bench._Gpu(0)
core, _ = bench.build_world(4096, 4, 64, 130., dev.device, seed=1, n_unique=1024)
print(bench.headline_env_step(dev, core, steps=100, warmup=10))
