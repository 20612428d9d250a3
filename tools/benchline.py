import sys, json
d = json.loads(sys.stdin.read())
print("ms_per_step", d["ms_per_step"])
for g in d["roofline"]["groups"]:
    if any(k in g["group"] for k in sys.argv[1:]):
        print(" ", g["group"], g["launches_per_step"], "x", round(g["avg_ms"] * 1e3, 1), "us")
