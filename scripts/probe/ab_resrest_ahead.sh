# A/B of the look-ahead depth of the walking residual+restriction kernel (level 1 of 512x512x64): bash scripts/probe/ab_resrest_ahead.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 11 12 13 21 31 32; do
  export MGX_RESREST_AHEAD=$v
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ahead_$v -- python3 $R/scripts/profile_solve.py 512 512 64 FC 10 > $R/gpurun_out/ahead_$v.log 2>&1 || exit 1
  echo "== MGX_RESREST_AHEAD=$v"
  python3 $R/scripts/solve_breakdown.py $(ls $R/gpurun_out/ahead_$v/*/*_kernel_trace.csv | head -1) 10 bygrid | grep -E "span|residual_restrict<"
done
