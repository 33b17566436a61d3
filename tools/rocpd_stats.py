"""Kernel statistics of a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) as CSV, demangled:
    python tools/rocpd_stats.py gpurun_out/prof2/st_results.db profiles/r01_kernel_stats_bench_steps20.csv"""
import csv, sqlite3, subprocess, sys      # noqa: E401


def main():
    db, out = sys.argv[1:3]
    rows = sqlite3.connect(db).execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r, n in zip(rows, names):
            w.writerow([n, r[1], round(r[2], 3), round(r[3], 3), round(r[4], 3)])
    print(f"{len(rows)} kernels -> {out}")


if __name__ == "__main__":
    main()
