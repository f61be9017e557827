# HBM traffic of the set-up kernels (two PMC passes, kernel trace only): bash scripts/probe/setup_pmc.sh   (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 250 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/su_fetch -- python3 $R/scripts/profile_setup.py 512 512 64 1 > $R/gpurun_out/su_fetch.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/su_write -- python3 $R/scripts/profile_setup.py 512 512 64 1 > $R/gpurun_out/su_write.log 2>&1 || exit 1
python3 - $(ls -t $R/gpurun_out/su_fetch/*/*_counter_collection.csv | head -1) $(ls -t $R/gpurun_out/su_write/*/*_counter_collection.csv | head -1) <<'PY'
import csv, sys
def top(path, counter):
    best = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        v = float(r["Counter_Value"])
        if v > best.get(k, 0.0): best[k] = v
    return best
f, w = top(sys.argv[1], "FETCH_SIZE"), top(sys.argv[2], "WRITE_SIZE")
print("largest (level-1) launch of each kernel: MB read (FETCH_SIZE KiB x 2, the gfx950 correction) / MB written")
for k in sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, 0) + w.get(k, 0))):
    if 2 * f.get(k, 0) + w.get(k, 0) > 20000: print("%-28s %8.0f MB read %8.0f MB written" % (k[:28], 2 * f.get(k, 0) * 1024 / 1e6, w.get(k, 0) * 1024 / 1e6))
PY
