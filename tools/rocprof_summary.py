#!/usr/bin/env python
"""Turn a rocprofv3 --kernel-trace --stats results db into a short text summary (kernel names truncated)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print(f"# {title}")
print(f"{'kernel':70s} {'calls':>8s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
for name, calls, total, avg, pct in rows[:8]:
    print(f"{name[:70]:70s} {calls:8d} {total:12.1f} {avg:10.3f} {pct:7.2f}")
