#!/usr/bin/env python3
"""How long the HOST spends in one pass of the sampler's hot loop (EnsembleMCMC.advance) at BASELINE
config 2, against the device time of the launch it queues: the host must stay ahead of the device
(it queues launch n + 1 while launch n runs), so its own time per pass bounds the whole-job rate.
    python tools/host_loop_probe.py [n_launches]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cobaya_amd.model import ProblemSpec  # noqa: E402
from cobaya_amd.sampler import MCMCHip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d = 30
mean, cov = bench.target(d)
info = bench.make_info(d, mean, cov, 65536, None, 40 * d)
s = MCMCHip(info["sampler"]["mcmc_hip"], ProblemSpec.from_info(info))
s._next_ckpt = s._checkpoint_steps()
for _ in range(60):
    s.advance()
s.engine.sync()
calls = []
t0 = time.perf_counter()
for _ in range(n):
    a = time.perf_counter()
    s.advance()
    calls.append(time.perf_counter() - a)
t_queue = time.perf_counter() - t0
s.engine.sync()
t_all = time.perf_counter() - t0
calls = np.array(calls) * 1e3
# the same loop with the device idle in between: the host's own cost of a pass
own = []
for _ in range(40):
    s.engine.sync()
    a = time.perf_counter()
    s.advance()
    own.append(time.perf_counter() - a)
own = np.array(own) * 1e3
print(f"{n} launches: queued in {1e3 * t_queue / n:.3f} ms per pass, done in {1e3 * t_all / n:.3f} ms per pass")
print(f"time inside advance() with the device busy: median {np.median(calls):.3f} ms, p90 {np.percentile(calls, 90):.3f}, max {calls.max():.3f}")
print(f"host's own cost of a pass (device idle before the call): median {np.median(own):.3f} ms, p90 {np.percentile(own, 90):.3f}, max {own.max():.3f} "
      f"(every 4th pass processes a checkpoint)")
s.close()
