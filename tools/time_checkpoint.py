"""Developer probe: where does a learn checkpoint spend its time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from cobaya_amd.model import ProblemSpec
from cobaya_amd.sampler import MCMCHip
d = 30
mean, cov = bench.target(d)
info = bench.make_info(d, mean, cov, 65536, None, 1200)
s = MCMCHip(info["sampler"]["mcmc_hip"], ProblemSpec.from_info(info))
eng = s.engine
for _ in range(3):
    eng.step(1200); eng.accumulate_moments()
eng.sync()
def T(f, n=5):
    t = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t) / n * 1e6
print("sync (idle)            %8.0f us" % T(eng.sync))
print("counters               %8.0f us" % T(eng.counters))
print("read_moments           %8.0f us" % T(lambda: eng.read_moments(reset=False)))
print("get_proposal_cov+set   %8.0f us" % T(lambda: eng.set_proposal_cov(eng.get_proposal_cov())))
for _ in range(2):
    eng.step(1200); eng.accumulate_moments()
    t = time.perf_counter(); s.check_convergence_and_learn_proposal(); s.i_learn += 1
    print("full checkpoint (incl. waiting for the queued launch) %8.0f us" % ((time.perf_counter() - t) * 1e6))
eng.sync()
t = time.perf_counter(); s.check_convergence_and_learn_proposal(); print("checkpoint, idle GPU   %8.0f us" % ((time.perf_counter() - t) * 1e6))
t = time.perf_counter(); eng.step(1200); print("step() call            %8.0f us" % ((time.perf_counter() - t) * 1e6))
t = time.perf_counter(); eng.accumulate_moments(); print("accumulate_moments()   %8.0f us" % ((time.perf_counter() - t) * 1e6))
t = time.perf_counter(); eng.sync(); print("sync after one launch  %8.0f us" % ((time.perf_counter() - t) * 1e6))
