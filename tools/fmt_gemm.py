import sys,json
for l in sys.stdin:
    if not l.startswith("{"): continue
    d=json.loads(l); print(d["shape"], d["epi"], " ".join(f"{k}={v}" for k,v in d.items() if k.endswith("_us")))
