import sys; sys.path.insert(0,".")
import numpy as np
from cobaya_amd.engine import Engine
d,W=27,65536
rng=np.random.default_rng(27)
A=rng.normal(size=(d,d)); s=10**rng.uniform(-2,np.log10(0.05),size=d)
c=A@A.T/d+np.eye(d); cov=c/np.sqrt(np.outer(np.diag(c),np.diag(c)))*np.outer(s,s); mean=np.full(d,0.5)
kinds=[0]*6+[1]*21; a=[0.0]*6+[0.5]*21; b=[1.0]*6+[0.3]*21
for inc in (False,True):
    eng=Engine(d,W,group_size=256,seed=1,incremental=inc,basis_group_size=4096 if inc else None)
    eng.set_prior(kinds,a,b); eng.set_target_gaussian(mean,cov); eng.set_proposal_cov(cov)
    eng.set_state(np.clip(mean+rng.standard_normal((W,d))*np.sqrt(np.diag(cov)),1e-6,1-1e-6))
    spl=40*d
    eng.step(spl); eng.sync(); eng.enable_timing(True); eng.kernel_times(reset=True)
    for _ in range(5): eng.step(spl)
    eng.sync(); kt=eng.kernel_times()
    print("config-5 shape", "incremental" if inc else "full", kt["step_ms"]/5, "ms per", spl, "->", W*spl*5/(kt["step_ms"]*1e-3), "evals/s;", eng.last_step_kernel(), "dirs", kt["basis_ms"]/5)
