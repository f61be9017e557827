# per-kernel time of a rebuild of all coefficients (nhydro_matrices): bash scripts/probe/setup_stats.sh [tag]   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=${1:-setup}
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T -- python3 $R/scripts/profile_setup.py 512 512 64 5 > $R/gpurun_out/$T.log 2>&1
grep "per rebuild" $R/gpurun_out/$T.log
python3 - $(ls -t $R/gpurun_out/$T/*/*_kernel_stats.csv | head -1) <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Name"].split("(")[0].replace("void ", "")
    t = int(r["TotalDurationNs"]) / 6e3
    tot += t
    if t > 30: print("%-34s calls/rebuild %5.1f avg %8.1f us  per rebuild %8.1f us" % (k[:34], int(r["Calls"]) / 6, float(r["AverageNs"]) / 1e3, t))
print("kernel time per rebuild %.0f us" % tot)
PY
