"""Worker of tests/test_cobaya_hosted.py: runs the REAL Cobaya of /root/reference
(`cobaya.run.run(info)`, `cobaya.output.load_samples`) with `sampler: mcmc_hip`, in its own
process so that importing Cobaya does not leak into the other tests.

The only thing replaced is the ctypes seam to libmcmc_hip.so: `_engine_factory` points at the
oracle-backed test double (tests/oracle_engine.py), because the build container has no GPU.
Everything above the seam is the product running under Cobaya's own `Sampler.__init__`:
initialize -> run -> checkpoints -> products -> resume.  Scenario `real_engine` leaves the
seam alone and checks that the un-faked run gets as far as `mcmc_hip_create`.

    python tests/_hosted_worker.py <scenario> <tmp dir>   -> one JSON line on stdout
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.dont_write_bytecode = True
sys.path[:0] = [ROOT, HERE, os.path.join(HERE, "golden", "_getdist_stub"), "/root/reference"]

import numpy as np  # noqa: E402

QUICK = {   # docs/src_examples/quickstart/gaussian.yaml of the reference (BASELINE config 1)
    "likelihood": {"gaussian_mixture": {"means": [0.2, 0], "covs": [[0.1, 0.05], [0.05, 0.2]],
                                        "derived": True}},
    "params": {"a": {"prior": {"min": -0.5, "max": 3}, "latex": r"\alpha"},
               "b": {"prior": {"dist": "norm", "loc": 0, "scale": 1}, "ref": 0,
                     "proposal": 0.5, "latex": r"\beta"},
               "derived_a": {"latex": r"\alpha^\prime"},
               "derived_b": {"latex": r"\beta^\prime"}},
}
TM = np.array([-0.48591462, 0.10064559, 0.64406749])     # tests/common_sampler.py:24-50
TC = np.array([[0.00078333, 0.00033134, -0.0002923],
               [0.00033134, 0.00218118, -0.00170728],
               [-0.0002923, -0.00170728, 0.00676922]])


def kl_norm(m1, S1, m2, S2):
    """KL(N1 || N2), cobaya/tools.py:732-743."""
    S2i = np.linalg.inv(S2)
    return float(0.5 * (np.trace(S2i @ S1) + (m1 - m2) @ S2i @ (m1 - m2) - len(m1)
                        + np.linalg.slogdet(S2)[1] - np.linalg.slogdet(S1)[1]))


def fake_seam():
    import mcmc_hip
    from oracle_engine import OracleEngine
    mcmc_hip.MCMCHip._engine_factory = staticmethod(OracleEngine)
    return mcmc_hip.MCMCHip


def quickstart(tmp):
    """Quickstart, every accepted row stored, output driver on."""
    from cobaya.output import load_samples
    from cobaya.run import run
    cls = fake_seam()
    info = dict(QUICK, output=os.path.join(tmp, "chains", "quick"),
                sampler={"mcmc_hip": {"seed": 3, "n_walkers": 256, "group_size": 64,
                                      "steps_per_launch": 50, "emit": "chains", "burn_in": 20,
                                      "max_samples": 120000, "Rminus1_stop": 0.0}})
    updated, sampler = run(info)
    from cobaya.collection import SampleCollection
    from cobaya.samplers.mcmc import MCMC
    prod = sampler.products()
    coll = prod["sample"]
    assert type(sampler) is cls and isinstance(sampler, MCMC)
    assert type(coll) is SampleCollection, type(coll)
    mean, cov = coll.mean(), coll.cov()
    tm, tc = np.array([0.2, 0.0]), np.array([[0.1, 0.05], [0.05, 0.2]])
    P = np.linalg.inv(tc) + np.diag([0.0, 1.0])      # x N(0,1) prior on b
    pc = np.linalg.inv(P)
    pm = pc @ np.linalg.inv(tc) @ tm
    files = sorted(os.listdir(os.path.join(tmp, "chains")))
    loaded = load_samples(os.path.join(tmp, "chains", "quick"), combined=True)
    upd_file = os.path.join(tmp, "chains", "quick.updated.yaml")
    from cobaya.yaml import yaml_load_file
    upd = yaml_load_file(upd_file)
    return {"columns": list(coll.data.columns), "n_rows": len(coll), "n": int(sampler.n()),
            "kl_like": kl_norm(tm, tc, mean, cov), "kl_post": kl_norm(pm, pc, mean, cov),
            "files": files, "loaded_rows": len(loaded),
            "loaded_mean": loaded.mean().tolist(), "mean": mean.tolist(),
            "progress_columns": list(prod["progress"].columns), "n_progress": len(prod["progress"]),
            "blocking": updated["sampler"]["mcmc_hip"]["blocking"],
            "updated_file_sampler": sorted(upd["sampler"]["mcmc_hip"]),
            "version": updated["sampler"]["mcmc_hip"].get("version"),
            "weights_int": bool(np.all(coll["weight"] == np.round(coll["weight"])))}


def fixed3(tmp):
    """tests/test_mcmc.py:22-82 / common_sampler.py:24-50,78-161: 3-d Gaussian, deliberately
    bad initial proposal, covariance learning on, run to convergence; no output driver."""
    from cobaya.run import run
    fake_seam()
    info = {
        "likelihood": {"gaussian_mixture": {"means": [TM], "covs": [TC],
                                            "input_params_prefix": "a_",
                                            "output_params_prefix": "", "derived": True}},
        "params": {**{f"a__{i}": {"prior": {"min": -1, "max": 1},
                                  "ref": {"dist": "norm", "loc": float(TM[i]), "scale": 0.2},
                                  "proposal": float(3 * np.sqrt(TC[i, i]))} for i in range(3)},
                   "_0": None, "_1": None, "_2": None},
        "sampler": {"mcmc_hip": {"seed": 11, "n_walkers": 512, "group_size": 64,
                                 "steps_per_launch": "20d", "max_tries": ".inf",
                                 "burn_in": "100d", "Rminus1_stop": 0.05,
                                 "max_samples": 2e7}},
    }
    updated, sampler = run(info)
    coll = sampler.products(skip_samples=0.5)["sample"]
    learned = sampler.proposer.get_covariance()
    prog = sampler.products()["progress"]
    return {"converged": bool(sampler.converged), "n_rows": len(coll),
            "kl": kl_norm(TM, TC, coll.mean(), coll.cov()),
            "learned_err": float(np.max(np.abs(learned - TC) / np.sqrt(np.outer(np.diag(TC),
                                                                                np.diag(TC))))),
            "Rminus1_last": float(prog["Rminus1"].iloc[-1]),
            "Rminus1_cl_last": float(prog["Rminus1_cl"].iloc[-1]),
            "derived_cols": [c for c in coll.data.columns if c.startswith("_")]}


def two_speeds(tmp):
    """Two likelihoods with speeds (the shape of tests/common_sampler.py:192-260): the blocks
    and oversampling factors come from Cobaya's live model through ProblemSpec."""
    from cobaya.run import run
    fake_seam()
    rng = np.random.default_rng(17)
    A = rng.normal(size=(3, 3))
    cov_b = (A @ A.T / 3 + np.eye(3)) * 0.01
    info = {
        "likelihood": {
            "slow": {"class": "gaussian_mixture", "means": [[0.2, 0.0]],
                     "covs": [[[0.1, 0.05], [0.05, 0.2]]], "input_params_prefix": "a_",
                     "speed": 1},
            "fast": {"class": "gaussian_mixture", "means": [[0.5, 0.4, 0.6]], "covs": [cov_b],
                     "input_params_prefix": "b_", "speed": 50}},
        "params": {**{f"a_{i}": {"prior": {"min": -3, "max": 3}, "ref": 0.1, "proposal": 0.3}
                      for i in range(2)},
                   **{f"b_{i}": {"prior": {"min": 0, "max": 1}, "ref": 0.5, "proposal": 0.1}
                      for i in range(3)}},
        "sampler": {"mcmc_hip": {"seed": 5, "n_walkers": 256, "group_size": 64,
                                 "oversample_power": 0.5, "steps_per_launch": "10d",
                                 "measure_speeds": False, "Rminus1_stop": 0.0,
                                 "max_samples": 150000, "snapshot_every": 40}}}
    updated, sampler = run(info)
    coll = sampler.products(skip_samples=0.3)["sample"]
    m, c = coll.mean(), coll.cov()
    tm = np.array([0.2, 0.0, 0.5, 0.4, 0.6])
    tc = np.zeros((5, 5))
    tc[:2, :2] = [[0.1, 0.05], [0.05, 0.2]]
    tc[2:, 2:] = cov_b
    return {"blocking": updated["sampler"]["mcmc_hip"]["blocking"],
            "cycle_length": int(sampler.cycle_length), "output_thin": int(sampler.output_thin),
            "kl": kl_norm(tm, tc, m, c), "columns": list(coll.data.columns),
            "chi2_sum_ok": bool(np.allclose(coll["chi2"], coll["chi2__slow"] + coll["chi2__fast"],
                                            rtol=1e-9, atol=1e-9))}


def one_nuisance(tmp):
    """The shape of a likelihood with ONE nuisance parameter (as plik-lite's A_planck): a slow
    block of several parameters and a fast block of one.  `evaluation: auto` must still pick
    the incremental path (one-parameter blocks draw the RandProposer1D variates there too)."""
    from cobaya.run import run
    fake_seam()
    info = {
        "likelihood": {
            "slow": {"class": "gaussian_mixture", "means": [[0.2, 0.0, 0.4]],
                     "covs": [[[0.1, 0.05, 0.0], [0.05, 0.2, 0.02], [0.0, 0.02, 0.05]]],
                     "input_params_prefix": "a_", "speed": 1},
            "fast": {"class": "gaussian_mixture", "means": [[1.0]], "covs": [[[0.0025]]],
                     "input_params_prefix": "cal_", "speed": 30}},
        "params": {**{f"a_{i}": {"prior": {"min": -3, "max": 3}, "ref": 0.1, "proposal": 0.3}
                      for i in range(3)},
                   "cal_0": {"prior": {"dist": "norm", "loc": 1.0, "scale": 0.1}, "ref": 1.0,
                             "proposal": 0.05}},
        "sampler": {"mcmc_hip": {"seed": 8, "n_walkers": 256, "group_size": 64,
                                 "oversample_power": 0.4, "steps_per_launch": "10d",
                                 "measure_speeds": False, "Rminus1_stop": 0.0,
                                 "max_samples": 120000, "snapshot_every": 30}}}
    updated, sampler = run(info)
    coll = sampler.products(skip_samples=0.3)["sample"]
    m, c = coll.mean(), coll.cov()
    tm = np.array([0.2, 0.0, 0.4, 1.0])
    tc = np.zeros((4, 4))
    tc[:3, :3] = [[0.1, 0.05, 0.0], [0.05, 0.2, 0.02], [0.0, 0.02, 0.05]]
    # likelihood N(1, 0.05^2) times prior N(1, 0.1^2) on the calibration parameter
    tc[3, 3] = 1.0 / (1.0 / 0.0025 + 1.0 / 0.01)
    return {"blocking": updated["sampler"]["mcmc_hip"]["blocking"],
            "incremental": bool(sampler.incremental), "kl": kl_norm(tm, tc, m, c),
            "cycle_length": int(sampler.cycle_length)}


def periodic_phase(tmp):
    """A `periodic: True` parameter (prior.py:658-676, mcmc.py `supports_periodic_params`) under
    the real Cobaya: read from the live model's prior, sampled on the incremental path."""
    from cobaya.run import run
    fake_seam()
    info = {
        "likelihood": {"gaussian_mixture": {"means": [[0.02, 0.5, 0.3]],
                                            "covs": [np.diag([0.0064, 0.004, 0.003]).tolist()],
                                            "input_params_prefix": "p_"}},
        "params": {"p_0": {"prior": {"min": 0, "max": 0.2}, "periodic": True, "ref": 0.05,
                           "proposal": 0.05},
                   "p_1": {"prior": {"min": 0, "max": 1}, "ref": 0.5, "proposal": 0.05},
                   "p_2": {"prior": {"min": 0, "max": 1}, "ref": 0.3, "proposal": 0.05}},
        "sampler": {"mcmc_hip": {"seed": 3, "n_walkers": 512, "group_size": 64,
                                 "Rminus1_stop": 0.0, "max_samples": 150000,
                                 "learn_every": "20d"}}}
    updated, sampler = run(info)
    x = sampler.engine.get_state()["x"]
    return {"incremental": bool(sampler.incremental),
            "periodic": [int(v) for v in sampler.spec.periodic],
            "inside": bool(np.all((x[:, 0] >= 0) & (x[:, 0] <= 0.2))),
            "high_end": int(np.sum(x[:, 0] > 0.15)), "low_end": int(np.sum(x[:, 0] < 0.05)),
            "mean0": float(x[:, 0].mean()), "mean1": float(x[:, 1].mean())}


def reference_test_mcmc(tmp):
    """The reference's own tests/test_mcmc.py::test_mcmc, with `mcmc` replaced by `mcmc_hip`:
    the SAME input -- `fixed_info` imported from the reference's tests/common_sampler.py, the
    deliberately bad 3 x 3 initial covmat, `max_tries 3000`, `burn_in 100 d`,
    `learn_proposal_Rminus1_max 30`, temperature 1 and 2, the check_gaussian callback -- and
    the same verdict: KL(truth || sample) <= KL_tolerance (0.07) on the later half of the
    sample (body_of_sampler_test, common_sampler.py:78-161; GetDist is absent here, so the
    sample moments come from Cobaya's own SampleCollection.mean / cov, detempered)."""
    sys.path.append("/root/reference")       # the reference's `tests` package (its fixtures)
    import importlib
    ref_cs = importlib.import_module("tests.common_sampler")
    from cobaya.run import run
    from cobaya.tools import KL_norm
    fake_seam()
    out = {}
    for temperature in (1, 2):
        cov = np.array([[0.01853538, -0.02990048, 0.00046138],
                        [-0.02990048, 0.14312571, -0.00441829],
                        [0.00046138, -0.00441829, 0.00019141]])
        seen = []

        def check_gaussian(sampler_instance):
            seen.append(float(KL_norm(
                S1=sampler_instance.model.likelihood["gaussian_mixture"].covs[0],
                S2=sampler_instance.proposer.get_covariance())))

        info = dict(ref_cs.fixed_info)
        dimension = 3
        info["sampler"] = {"mcmc_hip": {
            "max_tries": 3000, "burn_in": 100 * dimension, "covmat": cov,
            "covmat_params": list(info["params"])[:dimension],
            "learn_proposal_Rminus1_max": 30, "temperature": temperature,
            "callback_function": check_gaussian, "callback_every": 100,
            "seed": int(np.random.default_rng(1).integers(0, 2 ** 31)),
            # the ensemble: 512 walkers in 8 R-1 groups
            "n_walkers": 512, "group_size": 64, "steps_per_launch": "20d", "snapshot_every": 60}}
        info["debug"] = False
        info["output"] = os.path.join(tmp, f"out_chain_T{temperature}")
        updated, sampler = run(info)
        products = sampler.products(combined=True, skip_samples=0.5)
        sample = products["sample"]
        gm = info["likelihood"]["gaussian_mixture"]
        kl = float(KL_norm(m1=gm["means"][0], S1=gm["covs"][0], m2=sample.mean(),
                           S2=sample.cov()))     # (mean / cov detemper on the fly)
        out[f"T{temperature}"] = {
            "kl": kl, "tolerance": float(ref_cs.KL_tolerance), "converged": bool(sampler.converged),
            "n_rows": len(sample), "callbacks": len(seen),
            "kl_proposer_first": seen[0] if seen else None,
            "kl_proposer_last": seen[-1] if seen else None,
            "temperature_of_sample": float(sample.temperature)}
    return out


def two_speeds_drag(tmp, chains=False):
    """The same two-speed model with `drag: True` (tests/test_mcmc.py:129-143 of the reference):
    the slow block is proposed, the fast one dragged along -- through Cobaya's live model, on
    the incremental dragging path."""
    from cobaya.run import run
    fake_seam()
    rng = np.random.default_rng(17)
    A = rng.normal(size=(3, 3))
    cov_b = (A @ A.T / 3 + np.eye(3)) * 0.01
    info = {
        "likelihood": {
            "slow": {"class": "gaussian_mixture", "means": [[0.2, 0.0]],
                     "covs": [[[0.1, 0.05], [0.05, 0.2]]], "input_params_prefix": "a_",
                     "speed": 1},
            "fast": {"class": "gaussian_mixture", "means": [[0.5, 0.4, 0.6]], "covs": [cov_b],
                     "input_params_prefix": "b_", "speed": 50}},
        "params": {**{f"a_{i}": {"prior": {"min": -3, "max": 3}, "ref": 0.1, "proposal": 0.3}
                      for i in range(2)},
                   **{f"b_{i}": {"prior": {"min": 0, "max": 1}, "ref": 0.5, "proposal": 0.1}
                      for i in range(3)}},
        "sampler": {"mcmc_hip": {"seed": 6, "n_walkers": 256, "group_size": 64, "drag": True,
                                 "oversample_power": 0.5, "steps_per_launch": "10d",
                                 "measure_speeds": False, "Rminus1_stop": 0.0,
                                 "max_samples": 60000, "snapshot_every": 10}}}
    if chains:   # every accepted dragging step closes a weighted row (mcmc.py:656-668, 691-707)
        info["sampler"]["mcmc_hip"].update(emit="chains", snapshot_every=None, burn_in=5)
    updated, sampler = run(info)
    coll = sampler.products(skip_samples=0.3)["sample"]
    tm = np.array([0.2, 0.0, 0.5, 0.4, 0.6])
    tc = np.zeros((5, 5))
    tc[:2, :2] = [[0.1, 0.05], [0.05, 0.2]]
    tc[2:, 2:] = cov_b
    return {"drag": bool(sampler.drag), "interp": int(sampler.drag_interp_steps),
            "incremental": bool(sampler.incremental), "cycle_length": int(sampler.cycle_length),
            "blocking": updated["sampler"]["mcmc_hip"]["blocking"],
            "kl": kl_norm(tm, tc, coll.mean(), coll.cov()), "n_rows": len(coll),
            "max_weight": float(np.max(coll["weight"])), "accepted": int(sampler.n())}


def two_speeds_drag_chains(tmp):
    """tests/test_mcmc.py:132-171 (`test_mcmc_drag_results`) in the reference's own output mode:
    the dragging sampler's product is a weighted chain."""
    return two_speeds_drag(tmp, chains=True)


def resume(tmp):
    """Three legs on one prefix through cobaya.run: stop at max_samples; `resume: True` with a
    larger budget continues (rows of the first leg stay, new ones are appended); `resume:
    True` again with nothing left to do leaves every file untouched."""
    from cobaya.run import run
    fake_seam()
    prefix = os.path.join(tmp, "r", "leg")
    opts = {"seed": 21, "n_walkers": 128, "group_size": 64, "steps_per_launch": 40,
            "Rminus1_stop": 0.0, "learn_every": "20d", "snapshot_every": 40}

    def leg(max_samples, **kw):
        info = dict(QUICK, output=prefix, sampler={"mcmc_hip": dict(opts,
                                                                    max_samples=max_samples)},
                    **kw)
        return run(info)[1]

    def chain():
        return open(prefix + ".1.txt").read().splitlines()

    s1 = leg(20000)
    rows1, steps1, n1 = chain(), int(s1.n_steps_raw), int(s1.n())
    s2 = leg(40000, resume=True)
    rows2, steps2 = chain(), int(s2.n_steps_raw)
    coll2 = s2.products()["sample"]
    mtime = {f: os.path.getmtime(os.path.join(tmp, "r", f)) for f in os.listdir(os.path.join(tmp, "r"))
             if not f.endswith(".yaml") and not f.endswith(".lock")}
    s3 = leg(40000, resume=True)
    rows3 = chain()
    same_mtime = all(os.path.getmtime(os.path.join(tmp, "r", f)) == t for f, t in mtime.items())
    # a one-go run of the same total length ends in the same state (bit-identical resume)
    other = os.path.join(tmp, "r2", "leg")
    info = dict(QUICK, output=other, sampler={"mcmc_hip": dict(opts, max_samples=40000)})
    s4 = run(info)[1]
    st2, st4 = s2.engine.get_full_state(), s4.engine.get_full_state()
    try:
        leg(40000)          # neither resume nor force: Cobaya itself refuses
        refused = False
    except Exception as e:  # LoggedError
        refused = "resum" in str(e).lower() or "force" in str(e).lower()
    s5 = leg(20000, force=True)
    return {"rows1": len(rows1), "rows2": len(rows2), "rows3": len(rows3),
            "head_kept": rows2[:len(rows1)] == rows1, "steps1": steps1, "steps2": steps2,
            "n1": n1, "n2": int(s2.n()), "coll2": len(coll2),
            "untouched": rows3 == rows2 and same_mtime,
            "s3_steps": int(s3.n_steps_raw), "bit_identical": bool(
                np.array_equal(st2["x"], st4["x"]) and np.array_equal(st2["weight"], st4["weight"])
                and int(st2["step"]) == int(st4["step"])),
            "one_go_rows": len(open(other + ".1.txt").read().splitlines()),
            "refused": refused, "forced_rows": len(chain()), "forced_steps": int(s5.n_steps_raw)}


def real_engine(tmp):
    """Seam untouched: the run must get as far as `mcmc_hip_create` and fail there (no gfx950
    device in the build container) with Cobaya's LoggedError -- no CPU fallback."""
    from cobaya.log import LoggedError
    from cobaya.run import run
    info = dict(QUICK, sampler={"mcmc_hip": {"n_walkers": 128, "group_size": 64}})
    try:
        run(info)
    except LoggedError as e:
        return {"error": str(e)}
    return {"error": None}


def unsupported(tmp):
    from cobaya.log import LoggedError
    from cobaya.run import run
    fake_seam()
    out = {}
    bad = dict(QUICK, prior={"ext": "lambda a, b: -a**2"},
               sampler={"mcmc_hip": {"n_walkers": 128, "group_size": 64}})
    for name, info in (("external_prior", bad),
                       ("one_group", dict(QUICK, sampler={"mcmc_hip": {"n_walkers": 64,
                                                                       "group_size": 64}}))):
        try:
            run(info)
            out[name] = None
        except LoggedError as e:
            out[name] = str(e)
    return out


if __name__ == "__main__":
    import logging
    logging.disable(logging.INFO)
    res = globals()[sys.argv[1]](sys.argv[2])
    print("RESULT " + json.dumps(res))
