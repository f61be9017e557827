# A/B of the residual+restriction kernels: walk (k_residual_restrict) against one round trip (k_residual_restrict_flat, S rows per wave)
# per level of the 512x512x64 hierarchy.  bash scripts/probe/ab_resrest_flat.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "0 2" "2097152 2" "100000000 2" "100000000 4"; do
  set -- $v
  export MGX_RESREST_FLAT_MAX=$1 MGX_RESREST_FLAT_S=$2
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/flat_$1_$2 -- python3 $R/scripts/profile_solve.py 512 512 64 FC 10 > $R/gpurun_out/flat_$1_$2.log 2>&1 || exit 1
  echo "== MGX_RESREST_FLAT_MAX=$1 (largest fine level, in cells, that takes the flat kernel) MGX_RESREST_FLAT_S=$2"
  python3 $R/scripts/solve_breakdown.py $(ls $R/gpurun_out/flat_$1_$2/*/*_kernel_trace.csv | head -1) 10 bygrid | grep -E "span|residual_restrict"
done
