#!/bin/bash
# A/B of one environment switch on the bench workload: scripts/ab_bench.sh MGX_NO_KSP   (prints V-cycle ms, F-cycle it/s for both arms, twice)
V=$1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2; do
  for arm in off on; do
    if [ $arm = on ]; then export $V=1; else unset $V; fi
    python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python3 -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$V=$arm', 'vcycle_ms=%.4f' % j['ms_per_step'], 'fcycle_per_s=%.2f' % j['fcycle_iterations_per_sec'], 'sweep_ms=%.4f' % j['roofline']['sweep_ms'], 'res=%r' % j['residual_after'])"
  done
done
