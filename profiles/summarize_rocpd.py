"""Summarise a rocprofv3 (ROCm 7.x rocpd SQLite) kernel trace into a per-kernel stats table.
usage: python profiles/summarize_rocpd.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("""select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start),
                          max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count)
                          from kernels group by name order by 3 desc""").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# total kernel time {tot / 1e6:.3f} ms")
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'pct':>6s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} "
          f"{'lds':>6s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s}")
    for r in rows:
        print(f"{r[0][:70]:70s} {r[1]:6d} {r[2] / 1e6:10.3f} {100 * r[2] / tot:6.2f} {r[3] / 1e3:10.1f} {r[4] / 1e3:9.1f} "
              f"{r[5] / 1e3:9.1f} {r[6]:6d} {r[7]:5d} {r[8]:5d} {r[9]:5d}")


if __name__ == "__main__":
    main(sys.argv[1])
