"""Condense a rocprofv3 kernel_stats.csv: short kernel names, calls, avg/total us."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = {}
for r in rows:
    name = r["Name"]
    m = re.search(r"(hg_\w+)", name)
    if m: short = m.group(1) + (re.search(r"<(\d+)>", name).group(0) if re.search(r"hg_stream_kernel<(\d+)>", name) else "")
    elif "radix_sort" in name or "merge_sort" in name or "onesweep" in name or "histogram" in name: short = "rocprim sort*"
    elif "partition" in name or "lookback" in name or "init_" in name: short = "rocprim select*"
    else: short = name[:60]
    a = agg.setdefault(short, [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
print(f"{'kernel':44s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:44s} {c:7d} {t/1e6:10.3f} {t/c/1e3:10.2f} {100*t/tot:6.2f}")
