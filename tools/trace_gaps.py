import csv,sys,glob
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def short(n):
    for k in ("k_zero_call","k_first_hit","k_worklist","k_gather_one","k_combine_parts","k_occ","k_build","k_block","k_ws"):
        if k in n: return k
    return n[:30]
# last 40 kernels
seq=[(short(r['Kernel_Name']), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
# find call boundaries: k_zero_call
calls=[]; cur=[]
for s in seq:
    if s[0]=="k_zero_call" and cur: calls.append(cur); cur=[]
    cur.append(s)
calls.append(cur)
for c in calls[-6:]:
    if c[0][0]!="k_zero_call": continue
    t0=c[0][1]
    print(" | ".join(f"{n} +{(a-t0)/1e3:.1f}..{(b-t0)/1e3:.1f} ({(b-a)/1e3:.1f})" for n,a,b in c))
