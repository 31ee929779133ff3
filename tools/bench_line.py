import sys, json
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]
print(tag, "%.3e kmers/s" % d["value"], "%.0f GB/s" % r["achieved"], "frac %.3f" % r["frac"], "min %.3f med %.3f ms" % (r["min_kernel_ms"], r["median_kernel_ms"]), d["parity_vs_oracle"])
