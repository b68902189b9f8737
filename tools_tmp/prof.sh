cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
HICMI_BENCH_NO_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_p2 -o p2 -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/prof_p2.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_p2/**/*kernel_stats.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:26]:
    if 'p2' in r['Name'] or 'ins' in r['Name'] or 'arr' in r['Name']:
        print(r['Name'][:44].ljust(44), r['Calls'].rjust(7), r['TotalDurationNs'].rjust(12), r['AverageNs'].rjust(12), r['Percentage'])
PY
