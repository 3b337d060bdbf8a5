"""Compare two per-launch CSVs written under CANONSWAP_PROFILE_CSV (tools/gpu_diag.py): python tools/cmp_layers.py old.csv new.csv"""
import csv, re, sys, collections


def agg(path):
    out = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\d+", "#", r["label"])
        a = out.setdefault(k, [0.0, 0.0, 0])
        a[0] += float(r["ms"]); a[1] += float(r["gflop"]); a[2] += 1
    return out


a, b = agg(sys.argv[1]), agg(sys.argv[2])
print(f"{'label':28s} {'n':>3s} {'old ms':>8s} {'new ms':>8s} {'new/old':>8s} {'new TF/s':>9s}")
for k in sorted(a, key=lambda k: -a[k][0]):
    if k not in b:
        continue
    o, n = a[k], b[k]
    print(f"{k:28s} {o[2]:3d} {o[0]:8.3f} {n[0]:8.3f} {n[0] / o[0]:8.3f} {n[1] / n[0] if n[0] else 0:9.1f}")
