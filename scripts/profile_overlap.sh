#!/bin/bash
# Two full-size ranks on the one GPU of a test box (gloo rendezvous, peer-to-peer pushes): V-cycle rate with the exchange beside the interior
# sweep (MGX_OVERLAP=1) and with one stream (the default), and a kernel trace of rank 0 with queue ids (scripts/overlap_trace.py).
# usage: scripts/profile_overlap.sh <outdir> [nx ny nz nsmall]
set -e
OUT=$1; NX=${2:-512}; NY=${3:-512}; NZ=${4:-64}; NS=${5:-256}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp; export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 WORLD_SIZE=2
for mode in overlap onestream; do
  export MASTER_PORT=$((29570 + RANDOM % 200))
  if [ $mode = onestream ]; then export MGX_OVERLAP=0; else export MGX_OVERLAP=1; fi
  for r in 0 1; do
    RANK=$r LOCAL_RANK=$r timeout -k 10 200 python3 "$R/bench.py" --gpus 2 --backend gloo --size $NX $NY $NZ --nsmall $NS --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/${mode}_rank$r.log" 2>&1 &
  done
  wait
  python3 - "$OUT/${mode}_rank0.log" $mode <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:10s}: {d['ms_per_step']:.3f} ms per V-cycle (2 ranks of {d['config']['workload']}), exchanges {d['exchanges'][0]['per_vcycle']}, level-1 fill {d['exchanges'][0]['halo_fill_us']}")
PY
done
export MGX_OVERLAP=1
export MASTER_PORT=$((29570 + RANDOM % 200))
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -k 10 240 rocprofv3 --kernel-trace --stats -d "$OUT/rank$r" -o r$r --output-format csv -- \
    python3 "$R/bench.py" --gpus 2 --backend gloo --size $NX $NY $NZ --nsmall $NS --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/prof_rank$r.log" 2>&1 &
done
wait
python3 "$R/scripts/overlap_trace.py" $(ls "$OUT"/rank0/*kernel_trace.csv | head -1)
rm -f "$OUT"/rank*/*kernel_trace.csv
