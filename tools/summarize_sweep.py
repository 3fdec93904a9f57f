import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print("lanes", d["config"]["lanes_per_qp"], "A", d["config"]["agents_per_gpu"], "%.3e solves/s" % d["value"],
              "kernel_us %.2f" % (d["roofline"]["kernel_ms"] * 1e3), "frac %.3f" % d["roofline"]["frac"])
