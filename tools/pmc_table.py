"""per-kernel table of every counter found under <dir>/pmc_*/ (rocprofv3 counter_collection csv) + kernel-trace stats"""
import csv, glob, os, sys
d = sys.argv[1]
tab = {}
for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        e = tab.setdefault(k, {}).setdefault(r["Counter_Name"], [0, 0.0])
        e[0] += 1
        e[1] += float(r["Counter_Value"])
stats = {}
for f in glob.glob(os.path.join(d, "kt", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
        stats[k] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"]))
names = sorted({c for v in tab.values() for c in v})
print("kernel,calls,avg_us,pct," + ",".join(n + "_per_launch" for n in names))
for k in sorted(tab, key=lambda k: -stats.get(k, (0, 0, 0))[2]):
    s = stats.get(k, (0, 0.0, 0.0))
    print("\"%s\",%d,%.2f,%.2f," % (k, s[0], s[1], s[2]) + ",".join("%.1f" % (tab[k][n][1] / max(tab[k][n][0], 1)) if n in tab[k] else "" for n in names))
