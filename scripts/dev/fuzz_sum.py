import sys, json
bad=0; n=0; plans={}
for l in sys.stdin:
    if not l.startswith("{"):
        if "rror" in l or "ssert" in l: print(l[:200])
        continue
    d=json.loads(l); n+=1
    plans[d["plan"]]=plans.get(d["plan"],0)+1
    ok=all(d[k]["loss_ok"] and d[k]["bad_rows"]==0 and d[k]["dW_err"]<=d[k]["tol"] and d[k].get("bit_identical",True) for k in ("rep0","rep1","rep2")) and d["status"]==0
    if not ok: bad+=1; print("BAD", json.dumps(d)[:400])
print("cases", n, "bad", bad, "plans", plans)
