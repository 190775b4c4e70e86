"""Runs only the bulk statistics kernels (for rocprofv3 --pmc passes): python tools/bulk_only.py N K reps"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from graphistician_amd import _abi as abi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0, seed=1234)
adj, w = bench.build_models(abi, N, K, 1, 0)
for name, m in (("adj", adj), ("w", w)):
    m.set(abi.ST_C, truth["c"], np.int32)
    ms, nb = m.bench_bulk(data, reps)
    print("%s N=%d K=%d: %.3f ms/launch, %.0f GB/s, %.0f algorithmic bytes/launch" % (name, N, K, ms, nb / ms / 1e6, nb))
