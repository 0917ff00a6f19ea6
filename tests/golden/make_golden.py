#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the Python reference.

Runs only in the build container, where the reference tree is mounted read-only at
/root/reference (it never travels to the GPU box).  Nothing of the reference's source is
copied: only numeric inputs/outputs of its functions are stored, as small .npz files.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Versions that produced the committed fixtures: numpy 2.2.6, scipy 1.15.3, python 3.10.12,
reference cobaya v3.6.2 (no numba => scipy special_ortho_group fallback).  The third-party
`getdist` is absent here; a 6-name stand-in (tests/golden/_getdist_stub) satisfies the
imports in cobaya/collection.py:18-19 and likelihoods/base_classes/cmblikes.py:15 and is never called.

Fixture ids follow SURVEY.md §8c (G1..G9); G10 (blocked / oversampled / dragging chains), G11
(parameter-blocking decisions), G12 (detempering, reweighting) and G13 (the plik-lite
arithmetic, planck_pliklite.py:32-155, on a SYNTHETIC plik-lite-shaped data set: the Planck
data itself is not available offline) were added for §8f; G2 and G8
share one file (g2_g8_haar_chainstats.npz).  G3 (radial law) is a statistical test of the
oracle's own stream (tests/test_oracle_c.py), not a stored vector.
"""
import copy
import json
import logging
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("COBAYA_REFERENCE", "/root/reference")
sys.path[:0] = [os.path.join(HERE, "_getdist_stub"), REF]

import numpy as np  # noqa: E402

from cobaya.likelihoods.gaussian_mixture import (  # noqa: E402
    info_random_gaussian_mixture,
    random_cov,
)
from cobaya.model import get_model  # noqa: E402
from cobaya.sampler import get_sampler  # noqa: E402
from cobaya.samplers.mcmc.proposal import BlockedProposer  # noqa: E402

logging.disable(logging.CRITICAL)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", os.path.relpath(path), os.path.getsize(path), "bytes")


def rng_state_json(rng):
    st = copy.deepcopy(rng.bit_generator.state)
    return json.dumps(st, default=lambda o: int(o))


# ---------------------------------------------------------------------------- targets
def make_targets():
    """Targets of BASELINE configs 2/4: info_random_gaussian_mixture with
    default_rng(0), O_std in [0.01, 0.05], unit box; plus a 3-mode d=4 mixture."""
    out = {}
    for d in (30, 100):
        info = info_random_gaussian_mixture(
            ranges=[[0, 1]] * d, n_modes=1, input_params_prefix="a_", O_std_min=0.01,
            O_std_max=0.05, mpi_aware=False, random_state=np.random.default_rng(0),
            add_ref=True)
        gm = info["likelihood"]["gaussian_mixture"]
        out[f"mean_d{d}"] = np.array(gm["means"][0])
        out[f"cov_d{d}"] = np.array(gm["covs"][0])
    save("targets", **out)


# ---------------------------------------------------------------------------- G1
def g1_transforms():
    """a1: BlockedProposer.set_covariance -> transform[b] (proposal.py:226-260)."""
    out = {}
    rs = np.random.default_rng(11)
    cases = []
    for d in (2, 3, 30, 100):
        cov = random_cov([[0, 1]] * d, O_std_min=0.01, O_std_max=0.5, random_state=rs)
        cases.append((f"d{d}", cov, [list(range(d))]))
    cov5 = random_cov([[0, 1]] * 5, O_std_min=0.1, O_std_max=2.0, random_state=rs)
    cases.append(("d5_2blocks", cov5, [[3, 1], [0, 2, 4]]))
    cases.append(("d5_3blocks", cov5, [[4], [0, 1], [2, 3]]))
    for name, cov, blocks in cases:
        bp = BlockedProposer(blocks, np.random.default_rng(0),
                             oversampling_factors=np.ones(len(blocks), dtype=int))
        bp.set_covariance(cov)
        out[name + "_cov"] = cov
        out[name + "_blocks"] = np.array(sum(blocks, []))
        out[name + "_blocklens"] = np.array([len(b) for b in blocks])
        for i, t in enumerate(bp.transform):
            out[f"{name}_transform{i}"] = t
    save("g1_transforms", **out)


# ---------------------------------------------------------------------------- G4
def g4_prior():
    """a7/a6: Prior.logps_internal (prior.py:733-763) and reduce_periodic (658-676)."""
    params = {
        "u0": {"prior": {"min": -0.5, "max": 3.0}},
        "n0": {"prior": {"dist": "norm", "loc": 0.3, "scale": 1.7}},
        "u1": {"prior": {"min": 0.0, "max": 1.0}, "periodic": True},
        "n1": {"prior": {"dist": "norm", "loc": -2.0, "scale": 0.05}},
        "u2": {"prior": {"dist": "uniform", "loc": -1.0, "scale": 2.0}},
    }
    model = get_model({"likelihood": {"one": None}, "params": params})
    pr = model.prior
    rs = np.random.default_rng(4)
    pts = rs.normal(size=(96, 5)) * np.array([1.5, 2.0, 0.8, 0.1, 0.9]) + np.array(
        [1.0, 0.3, 0.5, -2.0, 0.0])
    # exact boundary points
    pts[0] = [-0.5, 0.0, 0.0, -2.0, -1.0]
    pts[1] = [3.0, 0.0, 1.0, -2.0, 1.0]
    pts[2] = [np.nextafter(3.0, 4.0), 0.0, 0.5, -2.0, 0.0]
    pts[3] = [1.0, 0.0, 0.5, -2.0, np.nextafter(-1.0, -2.0)]
    logp = np.array([pr.logps_internal(p) for p in pts])
    wrapped = np.array([pr.reduce_periodic(p, copy=True) for p in pts])
    save("g4_prior", points=pts, logprior=logp, wrapped=wrapped,
         bounds=np.array(pr.bounds()), kinds=np.array([0, 1, 0, 1, 0]),
         loc=np.array([0, 0.3, 0, -2.0, 0]), scale=np.array([0, 1.7, 0, 0.05, 0]),
         periodic=np.array([0, 0, 1, 0, 0]), uniform_logp=np.array(pr._uniform_logp))


# ---------------------------------------------------------------------------- G5
def g5_loglike():
    """a9/a10: GaussianMixture.logp (gaussian_mixture.py:138-163) incl. derived params;
    Gaussian.logp (gaussian.py:96-112) normalized on/off."""
    out = {}
    rs = np.random.default_rng(5)
    for d, K in ((2, 1), (3, 1), (3, 3), (4, 2), (30, 1), (30, 3), (100, 1)):
        info = info_random_gaussian_mixture(
            ranges=[[0, 1]] * d, n_modes=K, input_params_prefix="a_",
            output_params_prefix="b_", O_std_min=0.02, O_std_max=0.2, derived=True,
            mpi_aware=False, random_state=rs)
        if K > 1:
            w = rs.uniform(0.2, 1.0, size=K)
            info["likelihood"]["gaussian_mixture"]["weights"] = list(w / w.sum())
        model = get_model(info)
        gm = model.likelihood["gaussian_mixture"]
        means, covs = np.array(gm.means), np.array(gm.covs)
        pts = means[rs.integers(K, size=64)] + rs.normal(size=(64, d)) * np.sqrt(
            covs[0].diagonal()) * rs.uniform(0.2, 3.0, size=(64, 1))
        pts = np.clip(pts, 1e-9, 1 - 1e-9)
        ll = np.empty(64)
        der = np.empty((64, K * d))
        for i, p in enumerate(pts):
            res = model.logposterior(p)
            ll[i] = res.loglikes[0]
            der[i] = res.derived
        tag = f"gm_d{d}_K{K}"
        out[tag + "_means"], out[tag + "_covs"] = means, covs
        out[tag + "_weights"] = (np.atleast_1d(gm.weights) if K > 1 else np.array([1.0]))
        out[tag + "_points"], out[tag + "_loglike"], out[tag + "_derived"] = pts, ll, der
    for d, normalized in ((3, True), (27, True), (27, False)):
        cov = random_cov([[0, 1]] * d, O_std_min=0.05, O_std_max=0.5, random_state=rs)
        mean = rs.uniform(0.3, 0.7, size=d)
        info = {"likelihood": {"gaussian": {"mean": mean, "cov": cov,
                                            "normalized": normalized}},
                "params": {f"p{i}": {"prior": {"min": -10, "max": 10}} for i in range(d)}}
        model = get_model(info)
        pts = mean + rs.normal(size=(64, d)) * np.sqrt(cov.diagonal())
        ll = np.array([model.logposterior(p).loglikes[0] for p in pts])
        tag = f"gauss_d{d}_norm{int(normalized)}"
        out[tag + "_mean"], out[tag + "_cov"] = mean, cov
        out[tag + "_points"], out[tag + "_loglike"] = pts, ll
    save("g5_loglike", **out)


# ---------------------------------------------------------------------------- G6/G7
QUICKSTART = {
    "likelihood": {"gaussian_mixture": {"means": [0.2, 0],
                                        "covs": [[0.1, 0.05], [0.05, 0.2]],
                                        "derived": True}},
    "params": {"a": {"prior": {"min": -0.5, "max": 3}, "latex": r"\alpha"},
               "b": {"prior": {"dist": "norm", "loc": 0, "scale": 1}, "ref": 0,
                     "proposal": 0.5, "latex": r"\beta"},
               "derived_a": {"latex": r"\alpha^\prime"},
               "derived_b": {"latex": r"\beta^\prime"}},
}  # values of docs/src_examples/quickstart/gaussian.yaml (BASELINE config 1)

FIXED3 = {
    "likelihood": {"gaussian_mixture": {
        "means": [np.array([-0.48591462, 0.10064559, 0.64406749])],
        "covs": [np.array([[0.00078333, 0.00033134, -0.0002923],
                           [0.00033134, 0.00218118, -0.00170728],
                           [-0.0002923, -0.00170728, 0.00676922]])],
        "input_params_prefix": "a_", "output_params_prefix": "", "derived": True}},
    "params": {"a__0": {"prior": {"min": -1, "max": 1}},
               "a__1": {"prior": {"min": -1, "max": 1}},
               "a__2": {"prior": {"min": -1, "max": 1}},
               "_0": None, "_1": None, "_2": None},
}  # values of tests/common_sampler.py:24-50


def run_chain(info, sampler_opts):
    """Initialise the reference `mcmc` sampler, snapshot (rng state, initial point,
    initial proposal covariance), run it, and return everything needed to replay it."""
    info = copy.deepcopy(info)
    model = get_model(info)
    opts = {"measure_speeds": False, "Rminus1_stop": 0.0, "Rminus1_cl_stop": 0.0}
    opts.update(sampler_opts)
    sampler = get_sampler({"mcmc": opts}, model)
    sampled = list(model.parameterization.sampled_params())
    state0 = rng_state_json(sampler._rng)
    x0 = sampler.current_point.values.copy()
    lp0 = sampler.current_point.logpost
    cov0 = sampler.proposer.get_covariance()
    learned = []
    orig = sampler.proposer.set_covariance

    def spy(m):
        orig(m)
        learned.append(np.array(m, copy=True))

    sampler.proposer.set_covariance = spy
    sampler.run()
    data = sampler.collection.data
    cols = list(data.columns)
    prog = sampler.progress
    out = {
        "rng_state": np.array(state0),
        "x0": x0, "logpost0": np.array(lp0), "cov0": cov0,
        "columns": np.array(cols),
        "data": data.to_numpy(dtype=np.float64),
        "n_steps_raw": np.array(sampler.n_steps_raw),
        "final_x": sampler.current_point.values.copy(),
        "final_weight": np.array(sampler.current_point.weight),
        "learned_covs": np.array(learned) if learned else np.zeros((0, 1, 1)),
        "progress_N": prog["N"].to_numpy(dtype=np.float64),
        "progress_acc": prog["acceptance_rate"].to_numpy(dtype=np.float64),
        "progress_Rminus1": prog["Rminus1"].to_numpy(dtype=np.float64),
        "learn_every": np.array(sampler.learn_every.value),
        "max_tries": np.array(sampler.max_tries.value),
        "burn_in": np.array(sampler.burn_in.value),
        "temperature": np.array(float(sampler.temperature)),
        "proposal_scale": np.array(float(sampler.proposal_scale)),
        # blocking (mcmc.py:320-410); lengths vary, hence flat arrays + sizes
        "block_sizes": np.array([len(b) for b in sampler.blocks]),
        "block_params": np.array([sampled.index(p) for b in sampler.blocks for p in b]),
        "oversampling": np.array(sampler.oversampling_factors, dtype=int),
        "drag": np.array(bool(sampler.drag)),
        "drag_interp_steps": np.array(getattr(sampler, "drag_interp_steps", 0) or 0),
        "drag_last_slow": np.array(sampler.i_last_slow_block if sampler.drag else -1),
        "cycle_length": np.array(sampler.cycle_length),
        "output_thin": np.array(sampler.current_point.output_thin),
    }
    return out


def g6_traces():
    """a12-a16: full chain traces of the reference sampler (mcmc.py:451-1032)."""
    out = {}
    cases = {
        "quick_nolearn": (QUICKSTART, {"seed": 1, "max_samples": 400,
                                       "learn_proposal": False}),
        "quick_learn": (QUICKSTART, {"seed": 2, "max_samples": 600}),
        "fixed3_T1": (FIXED3, {"seed": 3, "max_samples": 500, "learn_proposal": False,
                               "burn_in": 20}),
        "fixed3_T2": (FIXED3, {"seed": 4, "max_samples": 500, "temperature": 2,
                               "learn_proposal": True}),
    }
    info30 = info_random_gaussian_mixture(
        ranges=[[0, 1]] * 30, n_modes=1, input_params_prefix="a_", O_std_min=0.01,
        O_std_max=0.05, mpi_aware=False, random_state=np.random.default_rng(0),
        add_ref=True)
    cov30 = np.array(info30["likelihood"]["gaussian_mixture"]["covs"][0])
    cases["d30_covmat"] = (info30, {
        "seed": 5, "max_samples": 300, "learn_proposal": False, "covmat": cov30,
        "covmat_params": list(info30["params"])})
    info4 = info_random_gaussian_mixture(
        ranges=[[0, 1]] * 4, n_modes=2, input_params_prefix="a_", O_std_min=0.02,
        O_std_max=0.08, mpi_aware=False, random_state=np.random.default_rng(7))
    info4["likelihood"]["gaussian_mixture"]["weights"] = [0.3, 0.7]
    cases["d4_K2"] = (info4, {"seed": 6, "max_samples": 400, "learn_proposal": True})
    for name, (info, opts) in cases.items():
        res = run_chain(info, opts)
        gm = info["likelihood"]["gaussian_mixture"]
        res["means"] = np.atleast_2d(np.array(gm["means"], dtype=float))
        covs = np.array(gm["covs"], dtype=float)
        res["covs"] = covs if covs.ndim == 3 else covs[None]
        res["weights"] = np.array(gm.get("weights") or [1.0], dtype=float)
        for k, v in res.items():
            out[f"{name}__{k}"] = v
        print(name, "rows", res["data"].shape, "steps", int(res["n_steps_raw"]))
    save("g6_traces", **out)


def g10_blocked():
    """(f)1: parameter blocks, oversampling, thinned output and dragging
    (proposal.py:96-260, mcmc.py:320-410, 545-668)."""
    info5 = info_random_gaussian_mixture(
        ranges=[[0, 1]] * 5, n_modes=1, input_params_prefix="a_", O_std_min=0.02,
        O_std_max=0.08, mpi_aware=False, random_state=np.random.default_rng(11),
        add_ref=True)
    n = list(info5["params"])
    cases = {
        "over_thin": {"seed": 21, "max_samples": 300, "learn_proposal": False,
                      "blocking": [[1, n[:2]], [3, n[2:]]]},
        "over_nothin": {"seed": 22, "max_samples": 400, "learn_proposal": True,
                        "oversample_thin": False, "blocking": [[1, n[3:]], [2, n[:3]]]},
        "blocks_1d": {"seed": 23, "max_samples": 300, "learn_proposal": False,
                      "oversample_thin": False,
                      "blocking": [[1, n[:3]], [2, n[3:4]], [4, n[4:]]]},
        "drag": {"seed": 24, "max_samples": 250, "learn_proposal": False, "drag": True,
                 "blocking": [[1, n[:2]], [4, n[2:]]]},
        "drag_learn": {"seed": 25, "max_samples": 300, "learn_proposal": True, "drag": True,
                       "burn_in": 5, "blocking": [[1, [n[4], n[0]]], [3, [n[1], n[3], n[2]]]]},
    }
    out = {}
    gm = info5["likelihood"]["gaussian_mixture"]
    out["means"] = np.atleast_2d(np.array(gm["means"], dtype=float))
    covs = np.array(gm["covs"], dtype=float)
    out["covs"] = covs if covs.ndim == 3 else covs[None]
    for name, opts in cases.items():
        res = run_chain(info5, opts)
        for k, v in res.items():
            out[f"{name}__{k}"] = v
        print(name, "rows", res["data"].shape, "steps", int(res["n_steps_raw"]),
              "blocks", res["block_sizes"], res["oversampling"], "drag",
              int(res["drag_interp_steps"]), "thin", int(res["output_thin"]))
    save("g10_blocked", **out)


def blocking_info(speeds):
    """Three Gaussian likelihoods over disjoint parameters with the given speeds."""
    return {
        "likelihood": {
            "slow": {"class": "gaussian_mixture", "means": [[0.2, 0]],
                     "covs": [[[0.1, 0.05], [0.05, 0.2]]], "input_params_prefix": "a_",
                     "speed": speeds[0]},
            "fast": {"class": "gaussian_mixture", "means": [[0.5, 0.5, 0.5]],
                     "covs": [(np.eye(3) * 0.01).tolist()], "input_params_prefix": "b_",
                     "speed": speeds[1]},
            "mid": {"class": "gaussian_mixture", "means": [[0.5]], "covs": [[[0.01]]],
                    "input_params_prefix": "c_", "speed": speeds[2]}},
        "params": {"a_0": {"prior": {"min": -3, "max": 3}}, "b_0": {"prior": {"min": 0, "max": 1}},
                   "a_1": {"prior": {"min": -3, "max": 3}}, "b_1": {"prior": {"min": 0, "max": 1}},
                   "c_0": {"prior": {"min": 0, "max": 1}}, "b_2": {"prior": {"min": 0, "max": 1}}}}


def g11_param_blocking():
    """(f)1: Model.get_param_blocking_for_sampler (model.py:1340-1467) decisions."""
    cases = []
    for speeds in ([1, 50, 7], [30, 2, 500], [5, 5, 5], [-1, 10, -1]):
        for power, split in ((0.0, False), (0.4, False), (1.0, False), (0.4, True), (0.7, True)):
            model = get_model(blocking_info(speeds))
            blocks, factors = model.get_param_blocking_for_sampler(
                split_fast_slow=split, oversample_power=power)
            cases.append({"speeds": speeds, "oversample_power": power, "split": split,
                          "blocks": [list(b) for b in blocks],
                          "factors": [int(f) for f in factors]})
    with open(os.path.join(HERE, "g11_param_blocking.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("wrote g11_param_blocking.json", len(cases), "cases")


def g12_detempering():
    """(f)4 collection utilities on a tempered chain (collection.py:688-763, 859-1019):
    detempered weights / -logpost, mean and covariance with and without tempering,
    reset_temperature, reweight."""
    info = copy.deepcopy(FIXED3)
    model = get_model(info)
    sampler = get_sampler({"mcmc": {"seed": 31, "max_samples": 400, "temperature": 3,
                                    "learn_proposal": False, "measure_speeds": False,
                                    "Rminus1_stop": 0.0, "Rminus1_cl_stop": 0.0}}, model)
    sampler.run()
    c = sampler.collection
    data = c.data.to_numpy(dtype=np.float64)
    out = {"columns": np.array(list(c.data.columns)), "data": data,
           "temperature": np.array(float(c.temperature)),
           "mean_tempered": c.mean(tempered=True), "cov_tempered": c.cov(tempered=True),
           "mean_detempered": c.mean(), "cov_detempered": c.cov(),
           "mean_slice": c.mean(first=50, last=300), "cov_slice": c.cov(first=50, last=300)}
    d1 = c.copy()
    d1.reset_temperature()
    out["reset_data"] = d1.data.to_numpy(dtype=np.float64)
    out["reset_temperature"] = np.array(float(d1.temperature))
    iw = np.random.default_rng(5).uniform(0, 2, size=len(c))
    iw[::7] = 0.0  # zero-weight rows are dropped
    d2 = c.copy()
    d2.reweight(iw.copy())
    out["importance_weights"] = iw
    out["reweight_data"] = d2.data.to_numpy(dtype=np.float64)
    save("g12_detempering", **out)
    print("g12 rows", data.shape, "->", out["reset_data"].shape, out["reweight_data"].shape)


def g2_g8_haar_and_chain_stats():
    """G2 (a3): PCG64 state -> SO(n) matrix of functions.random_SO_N (scipy fallback here);
    G8 (a15): SampleCollection.mean / cov over [first:last] of a stored integer-weight chain."""
    from cobaya.functions import random_SO_N
    out = {}
    for n in (2, 3, 5, 30):
        rng = np.random.default_rng(1000 + n)
        out[f"haar_state_{n}"] = np.array(rng_state_json(rng))
        out[f"haar_{n}"] = random_SO_N(n, random_state=rng)
        out[f"haar_next_normal_{n}"] = np.array(rng.standard_normal())  # stream position after
    model = get_model(copy.deepcopy(QUICKSTART))
    sampler = get_sampler({"mcmc": {"seed": 41, "max_samples": 500, "learn_proposal": False,
                                    "measure_speeds": False, "Rminus1_stop": 0.0,
                                    "Rminus1_cl_stop": 0.0}}, model)
    sampler.run()
    c = sampler.collection
    out["chain_columns"] = np.array(list(c.data.columns))
    out["chain_data"] = c.data.to_numpy(dtype=np.float64)
    for tag, (first, last) in {"all": (None, None), "mid": (100, 400), "tail": (250, None)}.items():
        out[f"mean_{tag}"] = c.mean(first=first, last=last)
        out[f"cov_{tag}"] = c.cov(first=first, last=last)
        out[f"mean_derived_{tag}"] = c.mean(first=first, last=last, derived=True)
        out[f"cov_derived_{tag}"] = c.cov(first=first, last=last, derived=True)
    save("g2_g8_haar_chainstats", **out)


def g7_multichain():
    """a16 multi-chain branch (mcmc.py:787-793, 856-889, 1021-1023) driven without MPI:
    m samplers in one process, `more_than_one_process` and `mpi.array_gather` patched so
    that sampler 0 sees the m chains (recipe of SURVEY Appendix C)."""
    import cobaya.mpi as cmpi
    import cobaya.samplers.mcmc.mcmc as mm

    m = 6
    samplers = []
    for i in range(m):
        model = get_model(copy.deepcopy(FIXED3))
        s = get_sampler({"mcmc": {"seed": 100 + i, "max_samples": 600,
                                  "covmat": FIXED3["likelihood"]["gaussian_mixture"][
                                      "covs"][0],
                                  "covmat_params": ["a__0", "a__1", "a__2"],
                                  "measure_speeds": False, "learn_proposal": False,
                                  "Rminus1_stop": 0.0, "Rminus1_cl_stop": 0.0}}, model)
        s.run()
        samplers.append(s)
    stats = []
    for s in samplers:
        first = int(s.n() / 2)
        stats.append([s.n(), s.collection.mean(first=first, tempered=True),
                      s.collection.cov(first=first, tempered=True),
                      s.get_acceptance_rate(first)])
    Ns = np.array([st[0] for st in stats], dtype=float)
    means = np.array([st[1] for st in stats])
    covs = np.array([st[2] for st in stats])
    accs = np.array([st[3] for st in stats])
    old_more, old_gather = mm.more_than_one_process, cmpi.array_gather
    mm.more_than_one_process = lambda: True
    cmpi.array_gather = lambda _lst: [Ns, means, covs, accs]
    mm.mpi.array_gather = cmpi.array_gather
    s0 = samplers[0]
    s0.learn_proposal = True
    s0.i_learn = 1
    try:
        s0.check_convergence_and_learn_proposal()
    finally:
        mm.more_than_one_process, cmpi.array_gather = old_more, old_gather
        mm.mpi.array_gather = old_gather
    chains = {}
    for i, s in enumerate(samplers):
        chains[f"chain{i}"] = s.collection.data.to_numpy(dtype=np.float64)
    save("g7_multichain", Ns=Ns, means=means, covs=covs, acceptance_rates=accs,
         Rminus1=np.array(s0.Rminus1_last),
         new_proposal_cov=s0.proposer.get_covariance(),
         progress_acc=np.array(float(s0.progress.at[1, "acceptance_rate"])),
         columns=np.array(list(samplers[0].collection.data.columns)), **chains)


# ---------------------------------------------------------------------------- G9
def g9_initial_covmat():
    """a17: CovmatSampler.initial_proposal_covmat precedence (sampler.py:485-685), with
    the construction of tests/test_mcmc_initial_covmat.py (dim 40)."""
    dim = 40
    rs = np.random.default_rng(9)
    i_s = list(range(dim))
    rs.shuffle(i_s)
    cov = random_cov(dim * [[0, 1]], random_state=rs)
    n_alt = dim // 4
    i_prop, i_ref, i_pri = i_s[:n_alt], i_s[n_alt:2 * n_alt], i_s[2 * n_alt:3 * n_alt]
    removed = i_prop + i_ref + i_pri
    i_cov = [i for i in range(dim) if i not in removed]
    for i in removed:
        diag = cov[i, i]
        cov[:, i] = 0
        cov[i, :] = 0
        cov[i, i] = diag
    order = list(range(dim))
    rs.shuffle(order)
    params = {}
    kind = np.zeros(dim, dtype=int)  # 0 covmat, 1 proposal, 2 ref, 3 prior
    for i in order:
        p = f"a_{i}"
        params[p] = {"prior": {"dist": "norm", "loc": 0, "scale": 1000}}
        sigma = np.sqrt(cov[i, i])
        if i in i_prop:
            params[p]["proposal"] = sigma
            kind[i] = 1
        elif i in i_ref:
            params[p]["ref"] = {"dist": "norm", "scale": sigma * 2}
            kind[i] = 2
        elif i in i_pri:
            params[p]["prior"]["scale"] = sigma * 2
            kind[i] = 3
    reduced = cov[np.ix_(i_cov, i_cov)]
    model = get_model({"likelihood": {"one": None}, "params": params})
    s = get_sampler({"mcmc": {"covmat": reduced,
                              "covmat_params": [f"a_{i}" for i in i_cov],
                              "measure_speeds": False}}, model)
    save("g9_initial_covmat", full_cov=cov, order=np.array(order), kind=kind,
         i_cov=np.array(i_cov), reduced=reduced,
         expected=cov[np.ix_(order, order)],
         got=s.proposer.get_covariance())


# ---------------------------------------------------------------------------- G13
class _Ini:
    """The five accessors `PlanckPlikLite.init_params` uses of getdist's IniFile
    (planck_pliklite.py:32-76), over a dict: this hands the reference its INPUT (key -> value,
    file names under a scratch directory); it implements nothing of the likelihood."""

    def __init__(self, folder, params):
        self.folder, self.params = folder, dict(params)

    def list(self, key):
        return str(self.params[key]).split()

    def int(self, key):
        return int(self.params[key])

    def int_list(self, key, default=None):
        return [int(x) for x in str(self.params[key]).split()] if key in self.params else default

    def string(self, key, default=None):
        return str(self.params.get(key, default))

    def relativeFileName(self, key):
        return os.path.join(self.folder, self.params[key])


def g13_pliklite():
    """§8f-4 / BASELINE configs[4]: `PlanckPlikLite.init_params` + `get_chi_squared`
    (planck_pliklite.py:32-155, `functions.chi_squared` 64-78) run on a synthetic data set with
    the layout of plik_lite_v22 (215 TT + 199 TE + 199 EE bins, l = 30..2508) written to a
    scratch directory in the reference's own file formats.  The REAL Planck data
    (plik_lite_2018_AL.zip) and a Boltzmann code are not available here: this pins the
    arithmetic, not the Planck numbers (tests/test_cosmo_planck_2018.py's chi2 = 584.24 stays
    unpinned)."""
    import tempfile

    from cobaya.likelihoods.base_classes.planck_pliklite import PlanckPlikLite

    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from cobaya_amd import pliklite as P  # the committed generator of the synthetic inputs

    ds = P.synthetic_dataset(seed=0)
    emu = P.synthetic_emulator(26, ds.lmax)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        np.savetxt(os.path.join(tmp, "data.txt"), ds.data, fmt="%.17g")
        np.savetxt(os.path.join(tmp, "blmin.txt"), ds.blmin, fmt="%d")
        np.savetxt(os.path.join(tmp, "blmax.txt"), ds.blmax, fmt="%d")
        np.savetxt(os.path.join(tmp, "weights.txt"), ds.weights, fmt="%.17g")
        np.savetxt(os.path.join(tmp, "cov.txt"), ds.cov, fmt="%.17g")
        base = {"nbintt": ds.nbintt, "nbinte": ds.nbinte, "nbinee": ds.nbinee, "lmax": ds.lmax,
                "bin_lmin_offset": ds.bin_lmin_offset, "data": "data.txt", "blmin": "blmin.txt",
                "blmax": "blmax.txt", "weights": "weights.txt", "cov_file": "cov.txt",
                "cov_file_binary": "absent.bin"}

        def reference_object(**over):
            like = object.__new__(PlanckPlikLite)   # no file loader, no installer, no provider
            like.init_params(_Ini(tmp, {**base, **over}))
            return like

        full = reference_object(use_cl="tt te ee")
        out.update(nbintt=ds.nbintt, nbinte=ds.nbinte, nbinee=ds.nbinee, lmax=ds.lmax,
                   bin_lmin_offset=ds.bin_lmin_offset, blmin=ds.blmin, blmax=ds.blmax,
                   weights_file=ds.weights, data=ds.data, cov=ds.cov.astype(np.float32),
                   ref_weights=full.weights, ref_blmin=full.blmin, ref_blmax=full.blmax,
                   ref_X_data=full.X_data, ref_used_indices=full.used_indices,
                   ref_invcov_diag=np.diag(full.invcov).copy(),
                   ref_invcov_row300=full.invcov[300].copy())
        assert np.array_equal(full.cov, ds.cov)     # the text round trip is exact
        # (a) 64 parameter draws of the committed linear emulator, a few sigma about the fiducial
        rs = np.random.default_rng(1313)
        bg = P.BinnedGaussian.from_dataset(ds)
        C = P.fisher_covariance(bg, emu)
        Lc = np.linalg.cholesky(C)
        z = rs.standard_normal((64, emu.n + 1)) * rs.choice([0.5, 1.0, 3.0], size=(64, 1))
        pts = z @ Lc.T
        theta, A = emu.theta0 + pts[:, :emu.n], 1.0 + pts[:, emu.n]
        A[:4] = [1.0, 0.99, 1.0025, 1.01]
        theta[0] = emu.theta0
        chi2 = np.empty(64)
        clsum = np.empty((64, 3))
        for k in range(64):
            D = emu.cl(theta[k])
            clsum[k] = D.sum(axis=1)
            chi2[k] = full.get_chi_squared(0, D[0], D[1], D[2], A[k])
        out.update(emu_theta=theta, emu_A=A, emu_chi2=chi2, emu_clsum=clsum)
        # (b) 8 explicit spectra (not from the emulator), float32-representable; L0 = 0 and 2
        D0 = emu.D0
        raw = (D0[None] * (1.0 + 0.05 * rs.standard_normal((8, 3, ds.lmax + 1)))).astype(np.float32)
        raw_A = np.array([1.0, 1.0, 0.995, 1.004, 1.0, 1.02, 0.98, 1.0])
        raw_L0 = np.array([0, 0, 0, 0, 2, 2, 2, 2])
        raw_chi2 = np.array([full.get_chi_squared(int(L0), *(r.astype(np.float64)[:, L0:]), A_planck=a)
                             for r, a, L0 in zip(raw, raw_A, raw_L0)])
        out.update(raw_cl=raw, raw_A=raw_A, raw_L0=raw_L0, raw_chi2=raw_chi2)
        # (c) bin / spectrum selections (planck_pliklite.py:84-125): TT only (the TT_lite_native
        # flavour), an explicit bin list, an l range
        for tag, over in (("tt", dict(use_cl="tt")),
                          ("bins", dict(use_cl="tt te ee", use_bins=" ".join(map(str, range(10, 120, 3))))),
                          ("lrange", dict(use_cl="te ee", bins_for_L_range="500 1200"))):
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):   # (the reference prints the l range)
                sub = reference_object(**over)
            out[f"sel_{tag}_used_indices"] = sub.used_indices
            out[f"sel_{tag}_chi2"] = np.array([
                sub.get_chi_squared(0, *emu.cl(theta[k]), A_planck=A[k]) for k in range(8)])
        # (d) the Fortran-binary covariance (planck_pliklite.py:60-67): ONE sequential
        # unformatted record of nbins^2 reals, written by scipy's FortranFile; only its LOWER
        # triangle is the covariance -- the upper one holds -1 here, which `np.tril` must drop
        from scipy.io import FortranFile
        junk = np.tril(ds.cov) + np.triu(np.full(ds.cov.shape, -1.0), 1)
        f = FortranFile(os.path.join(tmp, "cov.bin"), "w")
        f.write_record(junk)
        f.close()
        binl = reference_object(use_cl="tt te ee", cov_file_binary="cov.bin", cov_file="absent.txt")
        assert np.array_equal(binl.cov, ds.cov)
        out["bin_chi2"] = np.array([binl.get_chi_squared(0, *emu.cl(theta[k]), A_planck=A[k])
                                    for k in range(8)])
        assert np.array_equal(out["bin_chi2"], chi2[:8])
    save("g13_pliklite", **out)


if __name__ == "__main__":
    make_targets()
    g1_transforms()
    g4_prior()
    g5_loglike()
    g6_traces()
    g7_multichain()
    g9_initial_covmat()
    g10_blocked()
    g11_param_blocking()
    g12_detempering()
    g2_g8_haar_and_chain_stats()
    g13_pliklite()
