"""Re-run single fuzz cases (tests/fuzz_parity.py one_case) and say which field differs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import fuzz_parity as F
for args in ((5, 5, 4, 1, 154869791, (1, 2, 9), 0.5, True), (17, 17, 4, 1, 567335410, (1, 2, 9), 1.0, True)):
    for chunks in ((1,), (1, 1), (1, 2), (1, 2, 9), (3,), (12,)):
        a = list(args); a[5] = chunks
        print(args[:5], chunks, "->", F.one_case(*a), flush=True)
