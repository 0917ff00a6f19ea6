#!/usr/bin/env python3
"""Times the REFERENCE sampler (`cobaya.samplers.mcmc`, imported from /root/reference) on the
benchmark targets -- build container only: the reference never travels to the GPU box.

    python tools/time_reference.py [--procs 8] [--max-samples 4000]

Per dimension d in (2, 30, 100): single-mode gaussian_mixture from
info_random_gaussian_mixture(default_rng(0)) (the target of BASELINE configs 2/4), proposal
covariance = target covariance, seed 1, `measure_speeds: False`, no output; the rate is
n_steps_raw / wall of `sampler.run()` (set-up excluded).  `--procs N` runs N independent
processes (different seeds; no MPI here) and reports the sum.  Results go to BASELINE.md §2.
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("COBAYA_REFERENCE", "/root/reference")


def one(args):
    d, seed, max_samples, learn = args
    sys.dont_write_bytecode = True
    sys.path[:0] = [os.path.join(HERE, "..", "tests", "golden", "_getdist_stub"), REF]
    import logging

    import numpy as np
    logging.disable(logging.CRITICAL)
    from cobaya.likelihoods.gaussian_mixture import info_random_gaussian_mixture
    from cobaya.model import get_model
    from cobaya.sampler import get_sampler
    info = info_random_gaussian_mixture(
        ranges=[[0, 1]] * d, n_modes=1, input_params_prefix="a_", O_std_min=0.01,
        O_std_max=0.05, mpi_aware=False, random_state=np.random.default_rng(0), add_ref=True)
    cov = np.array(info["likelihood"]["gaussian_mixture"]["covs"][0])
    model = get_model(info)
    sampler = get_sampler({"mcmc": {
        "seed": seed, "max_samples": max_samples, "learn_proposal": learn,
        "measure_speeds": False, "Rminus1_stop": 0.0, "Rminus1_cl_stop": 0.0,
        "covmat": cov, "covmat_params": list(info["params"])}}, model)
    t0 = time.perf_counter()
    sampler.run()
    dt = time.perf_counter() - t0
    return sampler.n_steps_raw, dt, len(sampler.collection) / max(sampler.n_steps_raw, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--max-samples", type=int, default=4000)
    a = ap.parse_args()
    print(f"host: {os.cpu_count()} logical CPUs")
    for d in (2, 30, 100):
        for learn in (False, True):
            steps, dt, acc = one((d, 1, a.max_samples, learn))
            print(f"d={d:3d} learn_proposal={learn!s:5}: 1 process  {steps} steps in {dt:.2f} s "
                  f"= {steps / dt:8.0f} evals/s (acceptance {acc:.2f})")
        with mp.get_context("spawn").Pool(a.procs) as pool:
            t0 = time.perf_counter()
            res = pool.map(one, [(d, 10 + i, a.max_samples, False) for i in range(a.procs)])
            wall = time.perf_counter() - t0
        total = sum(r[0] for r in res)
        rate = sum(r[0] / r[1] for r in res)
        print(f"d={d:3d} {a.procs} independent processes: {total} steps, sum of per-process "
              f"rates {rate:8.0f} evals/s (pool wall {wall:.1f} s incl. start-up)")


if __name__ == "__main__":
    main()
