#!/usr/bin/env python3
"""Summarise a rocprofv3 result (rocpd sqlite .db, the default output of `rocprofv3 --kernel-trace
--stats`) as text for profiles/: per-kernel calls / total / average duration + launch resources."""
import sqlite3
import sys


def main(path, out=sys.stdout):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path, file=out)
    print("%-12s %8s %16s %16s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"), file=out)
    for name, calls, total, avg, pct in cur.execute("select * from top_kernels"):
        print("%s\n%-12s %8d %16.0f %16.0f %8.3f" % (name, "", calls, total, avg, pct), file=out)
    print("\n# dispatches (ns)", file=out)
    for name, start, end in cur.execute("select name, start, end from kernels order by start"):
        print("%12d  %s" % (end - start, name[:100]), file=out)
    print("\n# kernel symbols: LDS bytes, scratch bytes, SGPRs, VGPRs", file=out)
    for row in cur.execute("select kernel_name, group_segment_size, private_segment_size, sgpr_count, "
                           "arch_vgpr_count from kernel_symbols where kernel_name like '%milzma%'"):
        print("  %s lds=%s scratch=%s sgpr=%s vgpr=%s" % row, file=out)


if __name__ == "__main__":
    main(sys.argv[1])
