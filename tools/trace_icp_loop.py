import csv, re, sys, statistics
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        m=re.search(r"\bk_[a-z0-9_]+", r["Kernel_Name"]); rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:25]))
rows.sort()
# longest run of alternating k_icp_eval / k_icp_step
ev=[(s,e,n) for s,e,n in rows if n in ("k_icp_eval","k_icp_step")]
best=[];cur=[]
for x in ev:
    if cur and x[0]-cur[-1][1] > 200000: 
        if len(cur)>len(best): best=cur
        cur=[]
    cur.append(x)
if len(cur)>len(best): best=cur
d_eval=[e-s for s,e,n in best if n=="k_icp_eval"]; d_step=[e-s for s,e,n in best if n=="k_icp_step"]
gaps=[best[i+1][0]-best[i][1] for i in range(len(best)-1)]
print("run length", len(best), "eval us median", statistics.median(d_eval)/1e3, "step us median", statistics.median(d_step)/1e3, "gap us median", statistics.median(gaps)/1e3, "gap p90", sorted(gaps)[int(0.9*len(gaps))]/1e3, "total ms", (best[-1][1]-best[0][0])/1e6)
big=[g for g in gaps if g>5000]; print("gaps > 5us:", len(big), "sum ms", sum(big)/1e6)
