"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, no trace flags) into
profiles/rNN_pmc_hbm_traffic.json: HBM bytes per launch for every kernel.

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_hbm_traffic.json

Counters are in KiB per dispatch; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests as 64 bytes, MI355X_MICROARCH.md,
HBM section)."""
import csv, glob, json, os, sys      # noqa: E401
from collections import defaultdict


def read(dirname, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    dbs = glob.glob(os.path.join(dirname, "**", "*_results.db"), recursive=True)
    if dbs:                                   # rocprofv3 >= 7.0 default output: rocpd sqlite, view counters_collection
        import sqlite3
        for f in dbs:
            con = sqlite3.connect(f)
            for name, value in con.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
                tot[name] += float(value)
                cnt[name] += 1
        return tot, cnt
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *_results.db or *counter_collection.csv under {dirname}")
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                tot[row["Kernel_Name"]] += float(row["Counter_Value"])
                cnt[row["Kernel_Name"]] += 1
    return tot, cnt


def main():
    fdir, wdir, out = sys.argv[1:4]
    ft, fc = read(fdir, "FETCH_SIZE")
    wt, wc = read(wdir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(ft) | set(wt)):
        n = max(fc.get(k, 0), wc.get(k, 0))
        fb = 2.0 * 1024.0 * ft.get(k, 0.0) / max(fc.get(k, 1), 1)
        wb = 1024.0 * wt.get(k, 0.0) / max(wc.get(k, 1), 1)
        kernels[k] = {"launches": n, "fetch_bytes_per_launch_corrected": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace flags) over: python bench.py --steps 4 --warmup 2 "
            "--no-cpu-baseline; counters in KiB per dispatch; FETCH_SIZE doubled (gfx950 tallies 128-B requests as 64 B, "
            "MI355X_MICROARCH.md HBM section); per-launch averages over every dispatch of the kernel in the run")
    sig = None
    if len(sys.argv) > 4:          # the bench line of the same tile cache: its tile_signature ties `roofline.traffic` to these passes
        sig = json.load(open(sys.argv[4])).get("tile_signature")
    with open(out, "w") as fh:
        json.dump({"note": note, "tile_signature": sig, "kernels": kernels}, fh, indent=1)
    print(f"{len(kernels)} kernels -> {out}")


if __name__ == "__main__":
    main()
