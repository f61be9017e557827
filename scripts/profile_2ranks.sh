#!/bin/bash
# rocprofv3 kernel statistics of a 2-rank bench rehearsal with both ranks on the one GPU of a test box (gloo rendezvous,
# p2p halo transport).  Each rank is its own `rocprofv3 -- python3 bench.py` (no launcher in between).
# usage: scripts/profile_2ranks.sh <outdir> <nx> <ny> <nz> <nsmall>
set -e
OUT=$1; NX=$2; NY=$3; NZ=$4; NS=$5
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)   # absolute: the profiler runs from /tmp
cd /tmp; export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29561 WORLD_SIZE=2
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/rank$r" -o r$r --output-format csv -- \
    python3 "$R/bench.py" --gpus 2 --backend gloo --size $NX $NY $NZ --nsmall $NS --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/rank$r.log" 2>&1 &
done
wait
tail -1 "$OUT/rank0.log" | cut -c1-300
