#!/usr/bin/env python
"""Turn a rocprofv3 (ROCm 7.2 rocpd sqlite) result into the per-kernel stats table we commit under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r01/r01_results.db profiles/r01_c3_kernel_stats.md "<command line>"
"""
import sqlite3
import sys


def main():
    db_path, out_path = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out_path, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\n")
        if cmd:
            f.write(f"command: `{cmd}`\n\n")
        f.write("durations in microseconds (rocpd `top_kernels` view: calls, total, average, % of GPU kernel time)\n\n")
        f.write("| kernel | calls | total_us | avg_us | pct |\n|---|---:|---:|---:|---:|\n")
        for name, calls, total, avg, pct in rows:
            short = name.replace("lr::(anonymous namespace)::", "lr::")
            if len(short) > 90:
                short = short[:87] + "..."
            f.write(f"| `{short}` | {calls} | {total:.1f} | {avg:.3f} | {pct:.2f} |\n")
    print(out_path)


if __name__ == "__main__":
    main()
