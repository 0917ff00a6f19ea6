import sys, os, logging
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cobaya_amd import run
logging.basicConfig(level=logging.INFO)
tm = np.array([-0.48591462, 0.10064559, 0.64406749])
tc = np.array([[0.00078333, 0.00033134, -0.0002923],[0.00033134, 0.00218118, -0.00170728],[-0.0002923, -0.00170728, 0.00676922]])
info = {"likelihood": {"gaussian_mixture": {"means": [tm], "covs": [tc], "input_params_prefix": "a_", "output_params_prefix": "", "derived": True}},
  "params": {**{f"a__{i}": {"prior": {"min": -1, "max": 1}, "ref": {"dist": "norm", "loc": float(tm[i]), "scale": 0.2}, "proposal": float(3*np.sqrt(tc[i,i]))} for i in range(3)}, "_0": None, "_1": None, "_2": None},
  "sampler": {"mcmc_hip": {"seed": 11, "n_walkers": 1024, "group_size": 64, "steps_per_launch": "20d", "max_tries": ".inf", "burn_in": "100d", "Rminus1_stop": 0.002, "max_samples": 5e6}}}
u, s = run(info)
print(s.progress.to_string())
print(s.proposer.get_covariance()/tc)
