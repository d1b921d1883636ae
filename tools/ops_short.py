import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["op"], d["workload"][:30], round(d["ms"],3), round(d["GBps"]))
