"""Summarise a rocprofv3 (rocpd SQLite) kernel trace into the classic `--stats` table:
   python tools/rocpd_stats.py <results.db> [out.md]"""
import sqlite3
import sys


def main(db_path, out=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, cnt, tot, avg, mn, mx in rows:
        short = n if len(n) < 110 else n[:107] + "..."
        lines.append(f"| `{short}` | {cnt} | {tot/1e6:.3f} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*tot/total:.2f} |")
    txt = "\n".join(lines) + f"\n\ntotal kernel time: {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n"
    if out:
        open(out, "a").write(txt)
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:])
