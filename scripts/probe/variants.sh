# run the bench with each variant library found under scripts/probe/variants (on the GPU box: the repo there is a scratch copy)
cd $GRAFT_REPO_ROOT
cp mgroms_amd/libmgx.so /tmp/libmgx_base.so
for v in base $(ls scripts/probe/variants/ | sed 's/libmgx_//; s/.so//'); do
  if [ $v = base ]; then cp /tmp/libmgx_base.so mgroms_amd/libmgx.so; else cp scripts/probe/variants/libmgx_$v.so mgroms_amd/libmgx.so; fi
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/var_$v.log 2>&1
  python - $v <<PY
import json, sys
d=json.loads(open("gpurun_out/var_%s.log" % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step %.4f launch_us %.2f rb_ms %.4f rb_sweep %.4f fcycle %.1f" % (d["ms_per_step"], 1e3*d["roofline"]["launch_ms"], d["also_rb"]["ms_per_step"], d["also_rb"]["sweep_ms"], d["fcycle_iterations_per_sec"]))
PY
done
cp /tmp/libmgx_base.so mgroms_amd/libmgx.so
