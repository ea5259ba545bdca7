"""Print the kernels of the LAST E-step's statistics phase (everything from the end of its last chain launch on) from a rocprofv3
kernel-trace directory: start offset, duration, queue, grid, name."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_chain_ss' in r['Kernel_Name']]
last = idx[-1]
t0 = int(rows[last]['Start_Timestamp'])
print(f"last chain launch: dur {(int(rows[last]['End_Timestamp']) - t0) / 1e3:.1f} us")
for r in rows[last:]:
    s = int(r['Start_Timestamp']); e = int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  q {r.get('Queue_Id', '?'):>3}  grid {r['Grid_Size_X']:>8}x{r['Grid_Size_Y']:<3} {r['Kernel_Name'][:64]}")
