import sqlite3,sys,glob
for f in glob.glob(sys.argv[1]+'/**/*.db',recursive=True):
    c=sqlite3.connect(f)
    for r in c.execute("select name, count(*), avg(end-start)/1000.0, sum(end-start)/1e6 from kernels group by name order by 4 desc limit 16"): print("%-70s %6d %8.2f us %8.2f ms"%(r[0][:70],r[1],r[2],r[3]))
