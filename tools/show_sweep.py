import json, re, sys
txt = open(sys.argv[1]).read().split("== ")
for blk in txt[1:]:
    head = blk.split("\n")[0]
    m = re.search(r'\{.*\}', blk)
    if not m:
        print(head, "FAILED")
        continue
    j = json.loads(m.group(0))
    r = j["roofline"]
    print("%-32s ms %6.1f frac %.3f bwd us %5.1f fwd us %5.1f" % (head, j["ms_per_step"], r["frac"], r["avg_launch_us"], r["fwd_step"]["avg_launch_us"]))
