cd /tmp && export TMPDIR=/tmp
for nt in 0 1; do
export MGX_C2F_NT=$nt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c2f_nt$nt -- python3 $GRAFT_REPO_ROOT/scripts/profile_vcycle.py 512 512 64 FC 10 > $GRAFT_REPO_ROOT/gpurun_out/c2f_nt$nt.log 2>&1
echo NT=$nt; grep -E "coarse2fine" $GRAFT_REPO_ROOT/gpurun_out/c2f_nt$nt/*/*_kernel_stats.csv | cut -c1-200
done
