import json,sys
t=sys.stdin.read(); d=json.loads(t[t.index("{"):])
print(d["shape"]["relations"], {k:(round(v["avg_ms"],3), round(v.get("GBps",0))) for k,v in d["kernels"].items() if k.startswith("k_rgcn") or k=="k_fixup"})
