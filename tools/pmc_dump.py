"""Per-kernel sums of rocprofv3 PMC counters from a results .db (or csv) directory: python tools/pmc_dump.py <dir> [filter]"""
import collections, glob, os, sqlite3, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
c = sqlite3.connect(dbs[-1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
# rocpd schema: a view `counters_collection` (dispatch, kernel name, counter name, value) exists in ROCm 7
view = [t for t in tabs if t == "counters_collection"]
if not view:
    print("tables:", tabs); sys.exit(1)
v = view[0]
cols = [r[1] for r in c.execute(f"pragma table_info({v})")]
kn = "kernel_name" if "kernel_name" in cols else [x for x in cols if "name" in x][0]
cn = "counter_name" if "counter_name" in cols else [x for x in cols if "counter" in x and "name" in x][0]
cv = "value" if "value" in cols else "counter_value"
did = "dispatch_id" if "dispatch_id" in cols else cols[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for name, cnt, val, di in c.execute(f"select {kn},{cn},{cv},{did} from {v}"):
    if flt and flt not in name:
        continue
    agg[name[:60]][cnt] += val
    disp[name[:60]].add(di)
for k, m in agg.items():
    n = max(1, len(disp[k]))
    print(k, f"({n} dispatches)")
    for cnt, val in sorted(m.items()):
        print(f"    {cnt:32s} {val / n:16.1f} per dispatch")
