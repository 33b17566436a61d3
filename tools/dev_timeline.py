"""Developer aid: the device timeline of the replayed step from a rocprofv3 --kernel-trace run (rocpd sqlite): for every kernel, the mean
duration and the mean idle time between its end and the start of the next dispatch (and between the previous dispatch's end and its start), over the steady-state replays.
    rocprofv3 --kernel-trace -d gpurun_out/tl -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline
    python tools/dev_timeline.py gpurun_out/tl"""
import glob, os, sqlite3, sys
from collections import defaultdict
db = glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0]
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t.lower()][0]
cols = [r[1] for r in con.execute(f"pragma table_info({view})")]
print("view", view, cols, file=sys.stderr)
rows = con.execute(f"select name, start, end from {view} order by start").fetchall()
# steady state: the last 60 % of the dispatches
rows = rows[int(len(rows) * 0.4):]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
before, bcnt = defaultdict(float), defaultdict(int)
tot_d = tot_g = 0.0
for (n, s, e), (n2, s2, e2) in zip(rows, rows[1:]):
    g = s2 - e
    if g > 50000 or g < -50000:       # host-side pauses between replays
        continue
    k = n.split("(")[0][:70]
    dur[k] += e - s; gap[k] += g; cnt[k] += 1
    k2 = n2.split("(")[0][:70]
    before[k2] += g; bcnt[k2] += 1
    tot_d += e - s; tot_g += g
print(f"kernel time {tot_d / 1e6:.2f} ms, idle between dispatches {tot_g / 1e6:.2f} ms ({100 * tot_g / (tot_d + tot_g):.1f} %)")
for k in sorted(cnt, key=lambda k: -(dur[k] + gap[k]))[:40]:
    print(f"{cnt[k]:6d} x  dur {dur[k] / cnt[k] / 1e3:7.2f} us   idle after {gap[k] / cnt[k] / 1e3:6.2f} us   idle before {before[k] / max(bcnt[k], 1) / 1e3:6.2f} us   {k}")
