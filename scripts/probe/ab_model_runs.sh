cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 16 64; do
  export MGX_MODEL_KR=$v
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ts_$v -- python3 $R/scripts/profile_timestep.py 512 512 64 3 > $R/gpurun_out/ts_$v.log 2>&1
  echo "== MGX_MODEL_KR=$v"; grep "per time step" $R/gpurun_out/ts_$v.log
  python3 - $(ls -t $R/gpurun_out/ts_$v/*/*_kernel_stats.csv | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Name"].split("(")[0].replace("void ", "")
    if k.startswith(("k_rhs", "k_correct")): print("%-30s calls %4s avg %8.1f us" % (k[:30], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
