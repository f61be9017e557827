# nz = 128 hierarchy (BASELINE config 5's level-1 shape): which residual+restriction kernel for the 256x256x64 second level?
# bash scripts/probe/ab_nz128.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 2097152 4194304; do
  export MGX_RESREST_FLAT_MAX=$v
  timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/nz128_$v -- python3 $R/scripts/profile_vcycle.py 512 512 128 FC 10 > $R/gpurun_out/nz128_$v.log 2>&1 || exit 1
  echo "== MGX_RESREST_FLAT_MAX=$v"
  python3 - $(ls $R/gpurun_out/nz128_$v/*/*_kernel_trace.csv | head -1) <<'PY'
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
d = defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    d[(n, r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = 0
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if k[0].startswith("k_relax") or k[0].startswith("k_res") or k[0].startswith("k_coarse") or k[0].startswith("k_fine"):
        tot += sum(v)
        if sum(v) / 10 / 1e3 > 15: print("%9.1f us/cycle %5.1f calls %8.2f avg  %s" % (sum(v) / 10 / 1e3, len(v) / 10, sum(v) / len(v) / 1e3, k))
print("cycle kernels total %.1f us per V-cycle" % (tot / 10 / 1e3))
PY
done
