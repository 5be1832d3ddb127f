"""Print a compact per-kernel summary from a rocprofv3 --stats CSV directory (tools/kstats.py <dir>)."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), f"{int(r['TotalDurationNs'])/1e6:10.3f} ms", f"{float(r['AverageNs'])/1e3:10.1f} us", r['Percentage'])
