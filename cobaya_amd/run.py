"""`run(info) -> (updated_info, sampler)` for the inputs the mcmc_hip path covers: the shape
of cobaya.run.run (cobaya/run.py:30-183) without the parts that are out of scope (post-processing,
other samplers, the `.input/.updated.yaml` dumps); `output`, `resume` and `force` behave as in
the reference.  When Cobaya itself is installed,
use `cobaya.run.run` with `sampler: {mcmc_hip: ...}` instead -- the class registers through
the top-level `mcmc_hip` module."""
from __future__ import annotations

import copy

from .model import ProblemSpec, UnsupportedModel
from .sampler import HIP_DEFAULTS, MCMC_DEFAULTS, LoggedError, MCMCHip, log


def load_info(info_or_yaml):
    """dict, YAML string or path to a .yaml file (cobaya/input.py:148 load_info_overrides)."""
    if isinstance(info_or_yaml, dict):
        return copy.deepcopy(info_or_yaml)
    import os

    import yaml
    text = info_or_yaml
    if isinstance(text, str) and os.path.exists(text):
        with open(text, encoding="utf-8") as f:
            text = f.read()
    return yaml.safe_load(text)


def run(info_or_yaml, **overrides):
    info = load_info(info_or_yaml)
    info.update(overrides)
    samplers = info.get("sampler") or {}
    if len(samplers) != 1:
        raise LoggedError(log, "exactly one sampler block is expected, got %s", list(samplers))
    (name, opts), = samplers.items()
    if name not in ("mcmc_hip", "MCMCHip"):
        raise LoggedError(log, "this driver only runs `sampler: mcmc_hip` (got '%s'); other "
                               "samplers need Cobaya itself", name)
    try:
        spec = ProblemSpec.from_info(info)
    except UnsupportedModel as e:
        raise LoggedError(log, "mcmc_hip cannot sample this model: %s", str(e)) from e
    sampler = MCMCHip(opts or {}, spec, output=info.get("output"), name=name,
                      resume=bool(info.get("resume")), force=bool(info.get("force")))
    updated = copy.deepcopy(info)
    updated["sampler"] = {name: sampler.info()}
    sampler.run()
    return updated, sampler
