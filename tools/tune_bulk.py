"""Times the bulk statistics kernels alone (gg_bench_bulk) for the bench workload;
GG_BULK_RW overrides the rows-per-workgroup heuristic.  Usage: python tools/tune_bulk.py [N] [K]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from graphistician_amd import _abi as abi  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
truth = bench.planted_truth(N, K, seed=1234)
data = abi.DeviceData.synthetic_sbm(N, truth["c"], truth["p"], truth["p_self"], truth["mu"], truth["sigma"], 0.0, 1.0,
                                    seed=1234)
adj, w = bench.build_models(abi, N, K, 1, 0)
for name, m in (("adj", adj), ("w", w)):
    m.set(abi.ST_C, truth["c"], np.int32)
    ms, nb = m.bench_bulk(data, 50)
    print("%s RW=%s N=%d K=%d: %.2f us/launch, %.0f GB/s" % (name, os.environ.get("GG_BULK_RW", "auto"), N, K, ms * 1e3,
                                                           nb / ms / 1e6))
