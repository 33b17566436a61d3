"""Per-kernel average of one derived counter from a rocprofv3 --pmc run (rocpd sqlite):
    python tools/pmc_dump.py <dir> <COUNTER> [substring filter]"""
import glob, os, sqlite3, sys
from collections import defaultdict
d, counter = sys.argv[1:3]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
tot, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):
    for name, value in sqlite3.connect(f).execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        tot[name] += float(value); cnt[name] += 1
rows = sorted(((tot[k] / cnt[k], cnt[k], k) for k in tot if flt in k), key=lambda r: -r[1])
for avg, n, k in rows[:40]:
    print(f"{counter} {avg:10.3f}  n={n:5d}  {k[:110]}")
