"""Print the parts of a bench.py JSON line that matter when comparing runs."""
import json
import sys
d = json.load(open(sys.argv[1]))
print("value %.1f  ms/step %.4f  frac %.3f  forms %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("weight_model_sweep_forms")))
for c in d["config"].get("chains", []):
    print("  seed %d: %8.1f sweeps/s  %.1f us  changes/sweep %.1f  rounds/sweep %.2f  forms %s  bulk launch %.1f us" % (
        c["chain_seed"], c["value"], c["ms_per_step"] * 1e3, c["label_changes_per_sweep"], c["scan_rounds_per_sweep"],
        c["weight_model_sweep_forms"], c["bulk_launch_ms"] * 1e3))
if "other_configs" in d:
    print({k: (round(v.get("value", 0), 1) if isinstance(v, dict) else v) for k, v in d["other_configs"].items()})
if "cpu_baseline" in d:
    print("cpu", d["cpu_baseline"]["value"], "parity", d.get("parity_checked"))
if "sharded_chain" in d:
    print("sharded:", {k: d["sharded_chain"].get(k) for k in ("value", "n_gpus", "ms_per_step", "error")})
