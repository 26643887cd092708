"""Condense a profile_round.sh output directory into the text committed under profiles/."""
import csv, glob, collections, sys, json, os
out = sys.argv[1]
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-config4 --no-config3")
for l in open(os.path.join(out, "bench_trace.json")):
    if l.startswith("{"):
        d = json.loads(l)
        print("# bench line under the profiler: value %.1f %s, ms_per_step %.4f, roofline.kernel_ms_avg %.4f (hipEvents)" % (
            d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"]))
stats = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
for f in stats:
    print("## kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    print("%-70s %8s %14s %12s %12s %12s %7s" % ("Name", "Calls", "TotalNs", "AvgNs", "MinNs", "MaxNs", "Pct"))
    for r in rows:
        print("%-70s %8s %14s %12.0f %12s %12s %7s" % (r["Name"][:70], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
# steady-state average of the dominant kernel: last 40 dispatches of the trace (the profiled pass of bench.py)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("## kernel_trace.csv: mean duration of the LAST 40 dispatches per kernel (ns)")
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        tail = v[-40:]
        print("%-70s n=%5d last40_avg=%10.0f" % (k[:70], len(v), sum(tail) / len(tail)))
for name in ("fetch", "write"):
    for f in glob.glob(out + f"/{name}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print(f"## --pmc {name.upper()}_SIZE (mean per dispatch, raw counter value; rocprofv3 unit: KiB)")
        for k, v in agg.items():
            for c, x in v.items():
                print("%-70s %s mean=%.1f n=%d" % (k[:70], c, sum(x) / len(x), len(x)))
