cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
HICMI_BENCH_NO_TIMING=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_p2w8 -o p2 -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 > gpurun_out/prof_p2w8.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/prof_p2w8/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(rows[0].keys())
ks=[r for r in rows if ('k_p2' in r['Kernel_Name'] or 'k_ins' in r['Kernel_Name'] or 'k_arr' in r['Kernel_Name'])]
# last step only: take second half by time
ts=sorted(int(r['Start_Timestamp']) for r in ks)
mid=ts[len(ts)//2]
# find gap: part2 of warmup vs step: use largest gap in start times
gaps=[(ts[i+1]-ts[i],i) for i in range(len(ts)-1)]
g,i=max(gaps)
cut=ts[i+1]
ks=[r for r in ks if int(r['Start_Timestamp'])>=cut]
t0=min(int(r['Start_Timestamp']) for r in ks); t1=max(int(r['End_Timestamp']) for r in ks)
print('part2 kernels', len(ks), 'span ms', (t1-t0)/1e6, 'sum ms', sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in ks)/1e6)
q=collections.Counter(r['Queue_Id'] for r in ks)
print('queues', q)
# concurrency histogram
ev=[]
for r in ks:
    ev.append((int(r['Start_Timestamp']),1)); ev.append((int(r['End_Timestamp']),-1))
ev.sort()
cur=0; last=ev[0][0]; hist=collections.Counter()
for t,d in ev:
    hist[cur]+=t-last; last=t; cur+=d
tot=sum(hist.values())
print({k: round(v/tot,3) for k,v in sorted(hist.items())})
PY
