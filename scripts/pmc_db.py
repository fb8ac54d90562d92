"""Per-kernel averages out of rocprofv3's rocpd sqlite output: python scripts/pmc_db.py results.db [name filter]"""
import sqlite3
import sys


def main(path, like="%"):
    con = sqlite3.connect(path)
    tabs = {r[0].rsplit("_", 5)[0]: r[0] for r in con.execute("select name from sqlite_master where type='table'")}
    kd, ks = tabs["rocpd_kernel_dispatch"], tabs["rocpd_info_kernel_symbol"]
    pe, pi = tabs["rocpd_pmc_event"], tabs["rocpd_info_pmc"]
    for r in con.execute(f"select s.kernel_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id "
                         f"where s.kernel_name like ? group by 1", (like,)):
        print(f"{r[0][:80]:80s} launches {r[1]:4d} avg_ns {r[2]:.0f}")
    q = (f"select s.kernel_name, i.name, sum(e.value), count(distinct d.id) from {pe} e join {kd} d on e.event_id=d.event_id "
         f"join {ks} s on d.kernel_id=s.id join {pi} i on e.pmc_id=i.id where s.kernel_name like ? group by 1,2")
    for r in con.execute(q, (like,)):
        print(f"{r[0][:60]:60s} {r[1]:28s} per_launch {r[2] / r[3]:.5g}")


if __name__ == "__main__":
    main(*sys.argv[1:3])
