import os, sys, faulthandler
sys.path.insert(0, os.getcwd())
from tests.fuzz_parity import one_case
print(one_case(150, 70, 1, 1, 4242, (1, 2, 6)), flush=True)
