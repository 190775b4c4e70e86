import cProfile, pstats, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
from graphistician_amd import _abi as abi
from graphistician_amd.adjacency import SBMAdjacencyDistribution
from graphistician_amd import ais as AIS
N, R = 8192, 3
truth = bench.planted_truth(N, R, seed=99)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=99)
np.random.seed(5)
sbm = SBMAdjacencyDistribution(N, C=R, pi=10.0, tau0=1.0, tau1=1.0, seed=20)
AIS.estimate_marginal_likelihood(sbm, data, N_samples=1, B=6, steps_per_B=2)
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
AIS.estimate_marginal_likelihood(sbm, data, N_samples=1, B=100, steps_per_B=2)
print("AIS chain: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
