"""Shape sweep of the float-tolerance models (logistic / probit eigenmodel, latent-distance adjacency and weights): the
per-part engine-vs-oracle comparisons of their test files over awkward sizes (N around tile and wavefront edges, every
latent dimension).  Test infrastructure (checks against oracle/).

    python tests/fuzz_float_models.py [seed]
"""
import os
import sys
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_eigenmodel as TE  # noqa: E402
from tests import test_latent_distance as TL  # noqa: E402
from tests import test_latent_weights as TW  # noqa: E402


def main(seed):
    rng = np.random.RandomState(seed)
    sizes = [2, 3, 5, 31, 63, 64, 65, 127, 128, 129, 191, 255, 257, 300]
    bad = 0
    for it in range(40):
        N = int(rng.choice(sizes))
        D = int(rng.randint(1, 9))
        dim = int(rng.randint(1, 5))
        beta = float(rng.choice([1.0, 1.0, 0.4, 0.0]))
        dens = float(rng.choice([0.05, 0.3, 0.8]))
        for name, fn, args in (("logistic", TE.test_engine_parts_match_oracle, (N, D)),
                               ("probit", TE.test_probit_engine_parts_match_oracle, (N, D)),
                               ("logistic-tempered", TE.test_engine_tempered_sweeps_match_oracle, (N, D, beta)),
                               ("latent", TL.test_engine_parts_match_oracle, (N, dim, beta)),
                               ("latent-w", TW.test_engine_parts_match_oracle, (N, dim, beta, dens))):
            try:
                fn(*args)
            except Exception:
                bad += 1
                print("FAIL %s %s" % (name, args), flush=True)
                traceback.print_exc(limit=3)
    print("float-model shape sweep: %d failures" % bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 0) else 0)
