# A/B of the prolongation's run length (coarse levels per lane): bash scripts/probe/ab_c2f_kc.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 2 4 8 32; do
  export MGX_C2F_KC=$v
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kc_$v -- python3 $R/scripts/profile_solve.py 512 512 64 FC 10 > $R/gpurun_out/kc_$v.log 2>&1 || exit 1
  echo "== MGX_C2F_KC=$v"
  python3 $R/scripts/solve_breakdown.py $(ls $R/gpurun_out/kc_$v/*/*_kernel_trace.csv | head -1) 10 bygrid | grep -E "span|coarse2fine"
done
