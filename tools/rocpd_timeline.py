"""Start and end of the last dispatches in a rocprofv3 result database (rocpd sqlite), in microseconds from the
first one listed, with the idle time before each: where a step's time goes between its kernels.
usage: rocpd_timeline.py RESULTS.db [dispatches] [skip_from_end]"""
import sqlite3
import sys

from rocpd_kernel_stats import short


def main() -> int:
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    rows = rows[len(rows) - n - skip:len(rows) - skip]
    t0, prev = rows[0][1], rows[0][1]
    for name, s, e in rows:
        print(f"{(s - t0) / 1e3:9.2f} {(e - t0) / 1e3:9.2f}  gap {(s - prev) / 1e3:6.2f}  run {(e - s) / 1e3:6.2f}  {short(name)}")
        prev = e
    return 0


if __name__ == "__main__":
    sys.exit(main())
