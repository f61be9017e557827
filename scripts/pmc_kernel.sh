#!/bin/bash
# Hardware-counter passes over a few V-cycles (run on the GPU box): gpurun -- 'bash scripts/pmc_kernel.sh TAG "CTR1 CTR2" ["CTR3 ..."]'
# One rocprofv3 --pmc pass per quoted group (kernel trace only, no runtime trace domains), per-kernel averages of the largest
# dispatch of each kernel to gpurun_out/pmc_<TAG>.txt.
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
n=0
: > $OUT/pmc_${TAG}.txt
for grp in "$@"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$n -- python3 $ROOT/scripts/profile_vcycle.py 512 512 64 FC 2 > $OUT/pmc_${TAG}_$n.log 2>&1 || { tail -5 $OUT/pmc_${TAG}_$n.log; exit 1; }
  python3 - $(ls $OUT/pmc_${TAG}_$n/*/*_counter_collection.csv | head -1) >> $OUT/pmc_${TAG}.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
best = collections.defaultdict(dict)   # kernel -> counter -> (grid, [values])
for r in rows:
    k = r["Kernel_Name"]; c = r["Counter_Name"]; v = float(r["Counter_Value"]); g = int(r["Grid_Size"])
    cur = best[k].get(c)
    if cur is None or g > cur[0]: best[k][c] = (g, [v])
    elif g == cur[0]: cur[1].append(v)
for k in sorted(best):
    if not any(s in k for s in ("k_relax", "k_residual", "k_coarse2fine", "k_fine2coarse")): continue
    print(k[:90], " ".join(f"{c}={sum(v)/len(v):.4g}" for c, (g, v) in sorted(best[k].items())))
PY
done
cat $OUT/pmc_${TAG}.txt
