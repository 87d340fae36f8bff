"""Per-kernel time summary (calls, total, average, min, max in microseconds) from a rocprofv3 result database
(rocpd sqlite, the default output of `rocprofv3 --kernel-trace --stats`), written as CSV.
usage: rocpd_kernel_stats.py RESULTS.db [OUT.csv]"""
import csv
import re
import sqlite3
import sys


def short(name: str) -> str:
    """Kernel name without the anonymous namespace, the return type and the argument list."""
    if name.startswith("_Z"):  # a mangled name (template kernels arrive that way): kernel<first integer argument>
        m = re.search(r"\d+([a-z][a-z0-9_]*_kernel)(?:ILi(\d+)E)?", name)
        if m:
            return m.group(1) + (f"<{m.group(2)}>" if m.group(2) else "")
    name = name.replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i][:90]
    return name[:90]


def main() -> int:
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
        "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
    out.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "percent"])
    for name, n, tot, avg, mn, mx in rows:
        out.writerow([short(name), n, f"{tot / 1e3:.1f}", f"{avg / 1e3:.2f}", f"{mn / 1e3:.2f}", f"{mx / 1e3:.2f}",
                      f"{100.0 * tot / total:.2f}"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
