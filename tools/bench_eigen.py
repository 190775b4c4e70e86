"""Times the logistic eigenmodel sweep (BASELINE.json configs[2]: N=4096, D=8)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphistician_amd import _abi as abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rng = np.random.RandomState(1234)
F = rng.randn(N, D); lam = rng.randn(D); mu0 = -1.28 + rng.randn()
Mu = mu0 + (F * lam).dot(F.T); np.fill_diagonal(Mu, mu0)
A = (rng.rand(N, N) < 1 / (1 + np.exp(-Mu))).astype(np.uint8)
d = abi.DeviceData(A, None)
m = abi.Model(kind=abi.LOGISTIC_EIG, N=N, D=D, K=1, B=1, seed=1, sigma_mu0=1.0, sigma_F=1.0, sigma_lmbda=1.0, mu_mu_0=-1.28)
m.set(abi.ST_F, rng.randn(N, D)); m.set(abi.ST_LMBDA, rng.randn(D)); m.set(abi.ST_MU0, np.array([-1.0])); m.set(abi.ST_MU_LMBDA, np.zeros(D))
m.resample(d, 1)
for parts, name in ((abi.PART_OMEGA, "omega"), (abi.PART_F, "F"), (abi.PART_MU0 | abi.PART_LMBDA, "globals")):
    t0 = time.perf_counter(); m.resample_part(d, parts); print(name, "%.2f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter()
m.resample(d, steps)
el = time.perf_counter() - t0
print("N=%d D=%d: %.2f ms/sweep, %.1f sweeps/s, LL=%.3f" % (N, D, el / steps * 1e3, steps / el, m.log_likelihood(d)))
