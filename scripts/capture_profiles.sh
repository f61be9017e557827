#!/bin/bash
# Capture the rocprofv3 evidence behind the numbers of DESIGN.md / bench.py for one round (run on the GPU box):
#   gpurun -- 'bash scripts/capture_profiles.sh r02'
# Writes raw output under gpurun_out/<round>_*/ and the summaries that are tracked under profiles/.
# PMC counters are collected in passes of their own (FETCH_SIZE and WRITE_SIZE do not fit one pass), never together with
# the runtime trace domains.
set -o pipefail
R=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
stats() {  # $1 = tag, rest = program
  local tag=$1; shift
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${R}_${tag} -- "$@" > $OUT/${R}_${tag}.log 2>&1 || return 1
  cp $(ls $OUT/${R}_${tag}/*/*_kernel_stats.csv | head -1) $ROOT/profiles/${R}_${tag}_kernel_stats.csv
}
# 1. the driver's bench command (N=1 defaults)
stats bench python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 3 --no-live-traffic || exit 1
# 2. V-cycles only (no set-up noise in the averages)
stats vcycle python3 $ROOT/scripts/profile_vcycle.py 512 512 64 FC 20 || exit 1
# 3. BASELINE config 5's level-1 shape (nz = 128): sweeps, and V-cycles of the 512x512x128 and of the real 512x1024x128 block
stats nz128 python3 $ROOT/scripts/sweep_time.py 512 512 128 FC 10 || exit 1
stats nz128_vcycle python3 $ROOT/scripts/profile_vcycle.py 512 512 128 FC 10 || exit 1
stats config5_512x1024x128 python3 $ROOT/scripts/profile_vcycle.py 512 1024 128 FC 5 || exit 1
# 3b. solve_p iterations (F-cycle + residual): per-kernel time per iteration
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $OUT/${R}_solve -- python3 $ROOT/scripts/profile_solve.py 512 512 64 FC 10 > $OUT/${R}_solve.log 2>&1 || exit 1
python3 $ROOT/scripts/solve_breakdown.py $(ls $OUT/${R}_solve/*/*_kernel_trace.csv | head -1) 10 > $ROOT/profiles/${R}_solve_breakdown.txt || exit 1
# 4. HBM traffic of the dominant kernels, same bench command, two PMC passes
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${R}_pmc_fetch -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sweep-reps 3 --no-live-traffic > $OUT/${R}_pmc_fetch.log 2>&1 || exit 1
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${R}_pmc_write -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sweep-reps 3 --no-live-traffic > $OUT/${R}_pmc_write.log 2>&1 || exit 1
python3 $ROOT/scripts/pmc_summary.py $(ls $OUT/${R}_pmc_fetch/*/*_counter_collection.csv | head -1) $(ls $OUT/${R}_pmc_write/*/*_counter_collection.csv | head -1) 16777216 > $ROOT/profiles/${R}_pmc_traffic.json || exit 1
# 5. the reference's timer table (mg_tictoc format) of a 5-iteration solve, after an untimed warm-up (scripts/tictoc_table.py)
timeout -k 10 120 python3 $ROOT/scripts/tictoc_table.py $ROOT/profiles/${R}_tictoc_512x512x64_FC_5it.txt FC > $OUT/${R}_tictoc.log 2>&1 || exit 1
# 6. red-black (the reference default) in its three modes, and the per-kernel time of a solve_p iteration in the sequential order
timeout -k 10 200 python3 $ROOT/scripts/rb_modes_time.py 512 512 64 --json $ROOT/profiles/${R}_rb_modes_512.json > $OUT/${R}_rb_modes.log 2>&1 || exit 1
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $OUT/${R}_rbseq_solve -- python3 $ROOT/scripts/profile_solve.py 512 512 64 RB 5 > $OUT/${R}_rbseq_solve.log 2>&1 || exit 1
python3 $ROOT/scripts/solve_breakdown.py $(ls $OUT/${R}_rbseq_solve/*/*_kernel_trace.csv | head -1) 5 bygrid > $ROOT/profiles/${R}_rbseq_window_solve_breakdown.txt || exit 1
# ... and with the walk over the whole level instead of the windowed walk (the fallback of a weakly contracting matrix)
MGX_NO_RBSEQ_WINDOW=1 timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $OUT/${R}_rbseq_walk_solve -- python3 $ROOT/scripts/profile_solve.py 512 512 64 RB 5 > $OUT/${R}_rbseq_walk_solve.log 2>&1 || exit 1
python3 $ROOT/scripts/solve_breakdown.py $(ls $OUT/${R}_rbseq_walk_solve/*/*_kernel_trace.csv | head -1) 5 bygrid > $ROOT/profiles/${R}_rbseq_solve_breakdown.txt || exit 1
# 7. HBM traffic of the red-black kernels (two PMC passes over two solve_p iterations)
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${R}_pmc_rb_fetch -- python3 $ROOT/scripts/profile_solve.py 512 512 64 RB 2 > $OUT/${R}_pmc_rb_fetch.log 2>&1 || exit 1
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${R}_pmc_rb_write -- python3 $ROOT/scripts/profile_solve.py 512 512 64 RB 2 > $OUT/${R}_pmc_rb_write.log 2>&1 || exit 1
python3 $ROOT/scripts/pmc_summary.py $(ls $OUT/${R}_pmc_rb_fetch/*/*_counter_collection.csv | head -1) $(ls $OUT/${R}_pmc_rb_write/*/*_counter_collection.csv | head -1) 16777216 > $ROOT/profiles/${R}_pmc_traffic_rb_window.json || exit 1
rm -f $OUT/${R}_*/*/*_kernel_trace.csv $OUT/${R}_*/*/*_counter_collection.csv
mkdir -p $OUT/profiles_${R} && cp $ROOT/profiles/${R}_* $OUT/profiles_${R}/
ls -la $OUT/profiles_${R}/
