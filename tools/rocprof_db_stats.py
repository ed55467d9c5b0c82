import sqlite3,sys
c=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kt=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from {kt} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc limit {sys.argv[2] if len(sys.argv)>2 else 12}"
for r in c.execute(q): print(f"{r[0][:100]:100s} n={r[1]:5d} avg={r[2]:8.1f}us tot={r[3]:8.2f}ms")
