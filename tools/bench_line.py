import sys, json
tag = sys.argv[1] if len(sys.argv) > 1 else ""
lines = [ln for ln in sys.stdin.read().strip().splitlines() if ln.startswith("{")]
if not lines:
    print(tag, "NO OUTPUT")
    sys.exit(0)
d = json.loads(lines[-1])
r = d["roofline"]
s = d.get("sustained") or {}
print(tag, "%.3e kmers/s" % d["value"], "%.0f GB/s" % r["achieved"], "frac %.3f" % r["frac"],
      "min %.3f med %.3f ms" % (r["min_kernel_ms"], r["median_kernel_ms"]),
      ("sustained %.3f ms frac %.3f" % (s["ms_per_step"], s["frac"])) if s.get("frac") else "",
      "stream %.0f GB/s" % r.get("same_run_stream_read_GBps", 0), d["parity_vs_oracle"])
