"""Timeline of the LAST trace of a rocprofv3 --kernel-trace csv of tools/prof_gradient_only.py: python tools/trace_timeline.py CSV REPS"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'epsm' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(rows) // int(sys.argv[2])
t0 = int(rows[-n]['Start_Timestamp'])
for r in rows[-n:]:
    a, b = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%8.1f us +%7.1f  %s" % ((a - t0) / 1e3, (b - a) / 1e3, r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]))
print("%8.1f us end" % ((int(rows[-1]['End_Timestamp']) - t0) / 1e3))
