"""Print the kernel timeline of the last E-step from a rocprofv3 kernel trace directory (tools/ss_probe.sh)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/stats/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_chain_ss' in r['Kernel_Name']]
# start of the last E-step = first chain launch after the last non-chain kernel that precedes the last chain launch
last = idx[-1]
i = last
while i > 0 and 'k_chain_ss' in rows[i - 1]['Kernel_Name']:
    i -= 1
t0 = int(rows[i]['Start_Timestamp'])
for r in rows[i:]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  grid {r['Grid_Size_X']:>7}  {r['Kernel_Name'][:60]}")
