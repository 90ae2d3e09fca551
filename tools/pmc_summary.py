"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (sum over dispatches of the LAST step).
usage: python tools/pmc_summary.py gpurun_out/pmc/p1 [p2 p3 ...]"""
import csv, sys, collections, re
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "")[:48]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
dur = collections.defaultdict(float)
for prefix in sys.argv[1:]:
    with open(prefix + "_counter_collection.csv") as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add((prefix, r["Dispatch_Id"]))
names = sorted({c for k in tot for c in tot[k]})
print("kernel".ljust(48), "disp", *[n[-22:].rjust(23) for n in names])
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", tot[k].get("FETCH_SIZE", 0))):
    nd = len(cnt[k]) // max(1, len(sys.argv) - 1)
    print(k.ljust(48), str(nd).rjust(4), *[("%.4g" % tot[k].get(n, float("nan"))).rjust(23) for n in names])
