export TMPDIR=/tmp
O=gpurun_out/r3t
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 profiles/time_train_step.py > $O/log.txt 2>&1
cp $(find $O/prof -name "t_kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
rm -rf $O/prof
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3t/train_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); n=sum(int(r['Calls']) for r in rows)
print('total ms', tot/1e6, 'calls', n)
for r in rows[:45]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {int(r['Calls']):7d} calls {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:110]}")
PY
