"""Sweeps/s of the N=4096, K=16 headline chain under engine tuning knobs (one process per setting).
python tools/ahead_tune.py [N K steps]"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = sys.argv[1] if len(sys.argv) > 1 else "4096"
K = sys.argv[2] if len(sys.argv) > 2 else "16"
steps = sys.argv[3] if len(sys.argv) > 3 else "400"
for env in ({"GG_AHEAD": "0"}, {}, {"GG_RIDERS": "384"}, {"GG_RIDERS": "512"}, {"GG_RIDERS": "1024"}, {"GG_RIDERS": "4096"},
            {"GG_BENCH_STEPWISE": "1"}, {"GG_BENCH_STEPWISE": "1", "GG_RIDE_SPLIT": "1.0"}, {"GG_BENCH_STEPWISE": "1", "GG_RIDE_SPLIT": "0.0"}):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-extras", "--nodes", N, "--blocks", K,
                          "--steps", steps], env=e, capture_output=True, text=True)
    try:
        j = json.loads(out.stdout.strip().splitlines()[-1])
        print(env, "%.0f sweeps/s  %.1f us/sweep  in-situ %.0f GB/s over %.1f us launches" % (
            j["value"], j["ms_per_step"] * 1e3, j["roofline"]["achieved"], j["roofline"]["launch_ms"] * 1e3), flush=True)
    except Exception as exc:
        print(env, "FAILED", exc, out.stderr[-500:], flush=True)
