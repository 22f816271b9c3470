#!/usr/bin/env python3
import json, sys
d = json.load(open(sys.argv[1]))
print(d["config"]["workload"], "->", d["value"], d["unit"], "| gemv", d["roofline"]["achieved"] if d.get("roofline") else None, "GB/s | prefill", d["prefill"]["tokens_per_sec"])
for k in d["kernels"]:
    print("   %-34s x%-5g %9.2f us/step %8.2f us/launch %8s GB/s" % (k["name"], k["launches_per_step"], k["us_per_step"], k.get("us_per_launch", 0), k.get("GBps")))
if d["prefill"].get("kernels"):
    print("  prefill %d tokens: %.2f ms" % (d["prefill"]["tokens"], d["prefill"]["ms"]))
    for k in d["prefill"]["kernels"]:
        print("   %-34s x%-5d %9.3f ms %8s TFLOP/s" % (k["name"], k["launches"], k["ms"], k["TFLOPs"]))
