import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["roofline"]["frac"])
print({k:v for k,v in d["symmetric_option"].items() if k!="what"})
for r in d["config4_gemv"]: print(r["dtype"], r["path"][:40], round(r["gemv_ms"],3), round(r["roofline_frac"],4))
