#!/bin/bash
# Register / LDS / scratch use of the kernels in a built object: tools/scripts/kernel_regs.sh gg_model [name filter]
set -e
obj=$1; [ -f "$obj" ] || obj=/root/repo/graphistician_amd/_build/$1.o
tmp=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin "$obj" "$tmp/fat.bin"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$tmp/fat.bin" --output="$tmp/dev.co"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/dev.co" | python3 -c '
import sys, re
flt = sys.argv[1] if len(sys.argv) > 1 else ""
cur = {}
def flush():
    if cur.get("name") and flt in cur["name"]:
        print("%-90s vgpr %3s agpr %3s sgpr %3s spill v%s s%s scratch %s lds %s" % (cur["name"][:90], cur.get("vgpr_count"), cur.get("agpr_count"), cur.get("sgpr_count"), cur.get("vgpr_spill_count"), cur.get("sgpr_spill_count"), cur.get("private_segment_fixed_size"), cur.get("group_segment_fixed_size")))
for line in sys.stdin:
    m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)$", line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count" and cur.get("name"): flush(); cur.clear()
    if k in ("name",) and "symbol" not in cur: pass
    cur[k] = v
    if k == "wavefront_size": flush(); cur.clear()
' "${2:-}" | c++filt | sort
rm -rf "$tmp"
