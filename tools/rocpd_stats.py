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
    if len(sys.argv) > 3:          # optional markdown summary of the top 30
        total = sum(r[2] for r in rows)
        with open(sys.argv[3], "w") as fh:
            fh.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline   (MI355X, tools/profile_r03.sh)\n")
            fh.write("# the run = warm-up + capture + 20 timed + 10 train-only graph replays + the eager per-launch timing passes (3 bursts of 4, 5 in-step passes);\n")
            fh.write(f"# durations are per kernel launch.  Total kernel time {total / 1e3:.1f} ms.  Names are mangled: IDF16b = __bf16, IDF16_ = _Float16,\n")
            fh.write("# rocprofv3 durations carry ~3 us of per-dispatch floor (adam_tick_kernel: one thread, ~4.7 us here; 1.8 us per launch in a replayed graph).\n")
            fh.write("# conv_igemm template arguments = BM, BN, waves M, waves N, register stages, split-K groups, 1x1 fast path, normalise-on-load.\n\n")
            fh.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
            for r in sorted(rows, key=lambda r: -r[2])[:30]:
                fh.write(f"| `{r[0]}` | {r[1]} | {r[2] / 1e3:.2f} | {r[3]:.1f} | {r[4]:.2f} |\n")


if __name__ == "__main__":
    main()
