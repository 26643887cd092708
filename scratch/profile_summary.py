"""Condense a profile_round.sh output directory into the files committed under profiles/: the summary text (stdout), and beside
it <round>_bench_kernel_stats.csv (the very csv the summary quotes) and <round>_traffic.json (HBM bytes per launch of the
accumulate kernel from the two PMC passes)."""
import csv, glob, collections, sys, json, os, shutil
out = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "rNN"
traffic_only = "--traffic-only" in sys.argv
if not traffic_only:
    print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-config4 --no-config3 --no-config5")
    for l in open(os.path.join(out, "bench_trace.json")):
        if l.startswith("{"):
            d = json.loads(l)
            print("# bench line under the profiler: value %.1f %s, ms_per_step %.4f, roofline.kernel_ms_avg %.4f (hipEvents)" % (
                d["value"], d["unit"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"]))
    print("# READ THIS FIRST (r4): a thread's downsample calls rotate over up to three workspaces and streams, so in the timed region of the bench")
    print("# (a stream of 200 calls) up to three accumulate kernels are in flight at once.  A kernel's OWN duration in that part of the trace is")
    print("# longer than alone (its workgroups share the chip with its neighbours') while calls COMPLETE faster: the rate is the start-to-start")
    print("# interval.  The csv's average mixes that part with the bench's second pass (every call waited for, the kernel alone).  Both parts are")
    print("# given below, and the same run with ONE workspace (nothing overlaps) is in %s_bench_kernel_stats_one_workspace.csv." % rnd)
    one = os.path.join(out, rnd + "_bench_kernel_stats_one_workspace.csv")
    if os.path.exists(one):
        for r in csv.DictReader(open(one)):
            if "voxel_accumulate" in r["Name"] and "general" not in r["Name"] and "<1" in r["Name"]:
                print("# one workspace (CWIPC_WORKSPACES=1): accumulate kernel %s calls, average %.0f ns, min %s ns" % (r["Calls"], float(r["AverageNs"]), r["MinNs"]))
        for l in open(os.path.join(out, "bench_trace_one.json")):
            if l.startswith("{"):
                d1 = json.loads(l)
                print("# one workspace, bench line under the profiler: ms_per_step %.4f" % d1["ms_per_step"])
    stats = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
    for f in stats:
        shutil.copyfile(f, os.path.join(out, rnd + "_bench_kernel_stats.csv"))
        print("## kernel_stats.csv (committed beside this file as %s_bench_kernel_stats.csv)" % rnd)
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
        # r4: the accumulate kernel in the two parts of the run.  In the timed region (a stream of calls on three streams) consecutive
        # accumulate kernels overlap on the chip: a kernel's own duration is LONGER than alone while calls complete FASTER -- the rate is
        # the start-to-start interval; in the second pass every call is waited for and the kernel runs alone.
        k1 = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))
                    if "voxel_accumulate" in r["Kernel_Name"] and "general" not in r["Kernel_Name"] and "kernel<0" not in r["Kernel_Name"])
        # the headline's kernel (<1, ...>) in the order bench.py launches it: 36 calls of the set-up burst (second session of r4), 20 warm-up
        # steps, 4 sizing calls, the 200 timed steps, the 200 steps of the second pass, then the call-then-count calls
        if len(k1) >= 460:
            timed = k1[60:260]
            alone = k1[260:460]
            print("## accumulate kernel, timed region (stream of calls): start-to-start %.0f ns, own duration %.0f ns (neighbours overlap)" % (
                (timed[-1][0] - timed[0][0]) / (len(timed) - 1), sum(e - s for s, e in timed) / len(timed)))
            print("## accumulate kernel, second pass (every call waited for, the kernel alone): duration %.0f ns, min %d" % (
                sum(e - s for s, e in alone) / len(alone), min(e - s for s, e in alone)))
    for name in ("fetch", "write"):
        for f in glob.glob(out + f"/{name}/**/*counter_collection.csv", recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for row in csv.DictReader(open(f)):
                agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
            print(f"## --pmc {name.upper()}_SIZE (mean per dispatch, raw counter value; rocprofv3 unit: KiB)")
            for k, v in agg.items():
                for c, x in v.items():
                    print("%-70s %s mean=%.1f n=%d" % (k[:70], c, sum(x) / len(x), len(x)))

# HBM bytes per launch of the accumulate kernel (what bench.py quotes as roofline.traffic)
means = {}
symbol = None
for name in ("fetch", "write"):
    for f in glob.glob(out + f"/{name}/**/*counter_collection.csv", recursive=True):
        vals = []
        for row in csv.DictReader(open(f)):
            if "voxel_accumulate" in row["Kernel_Name"] and "general" not in row["Kernel_Name"] and row["Counter_Name"] == name.upper() + "_SIZE":
                vals.append(float(row["Counter_Value"]))
                symbol = row["Kernel_Name"].split("(")[0].split("::")[-1]
        if vals:
            means[name] = sum(vals) / len(vals)
if "fetch" in means and "write" in means:
    n_points = None
    for l in open(os.path.join(out, "bench_fetch.json")):
        if l.startswith("{"):
            n_points = json.loads(l)["config"]["points_per_gpu"]
    rec = {
        "round": int(rnd.lstrip("r") or 0),
        "command": "rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-config4 --no-config3 --no-config5 ; same with --pmc WRITE_SIZE (separate passes; scratch/profile_round.sh %s)" % rnd,
        "workload_points": n_points,
        "kernel": "voxel_accumulate",
        "kernel_symbol": symbol,
        "FETCH_SIZE_KiB_raw_mean_per_dispatch": means["fetch"],
        "WRITE_SIZE_KiB_mean_per_dispatch": means["write"],
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request for 16 B/lane streaming loads -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE taken as is (64-bit atomics on 64-B records: uncalibrated access width)",
        "hbm_bytes_per_launch": int(round((2 * means["fetch"] + means["write"]) * 1024)),
    }
    json.dump(rec, open(os.path.join(out, rnd + "_traffic.json"), "w"), indent=1)
    print("## traffic: %d B per launch (2 x FETCH_SIZE + WRITE_SIZE) -> %s_traffic.json" % (rec["hbm_bytes_per_launch"], rnd))
