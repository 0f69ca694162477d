import sqlite3,sys,collections
db=sqlite3.connect(sys.argv[1]); pat=sys.argv[2]
cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
v=[t for t in tabs if t.startswith('counters_collection')]
cols=[r[1] for r in cur.execute(f"pragma table_info({v[0]})")]
print(cols)
acc=collections.defaultdict(list)
for row in cur.execute(f"select * from {v[0]}"):
    r=dict(zip(cols,row))
    if pat in str(r.get('kernel_name','')): acc[r['counter_name']].append(r['value'])
for k,vals in acc.items(): print(k,len(vals),sum(vals)/len(vals))
