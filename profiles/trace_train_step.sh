export TMPDIR=/tmp
O=gpurun_out/r4t
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 profiles/time_train_step.py > $O/log.txt 2>&1
cp $(find $O/prof -name "t_kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
rm -rf $O/prof
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4t/train_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); n=sum(int(r['Calls']) for r in rows)
nat=sum(int(r['Calls']) for r in rows if 'at::native' in r['Name'] or 'rocclr' in r['Name'])
print('total ms', tot/1e6, 'calls', n, 'per step (7 steps):', n/7, 'launches', tot/7e6, 'ms; at::native + copy share of launches', nat/n)
for r in rows[:50]:
    print(f"{float(r['TotalDurationNs'])/7e3:9.1f} us/step {int(r['Calls'])/7:7.1f} calls/step {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
tail -1 $O/log.txt | cut -c1-300
