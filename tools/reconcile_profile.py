"""Reconcile a rocprofv3 --kernel-trace --stats summary with a bench.py JSON line.
    python tools/reconcile_profile.py profiles/X_kernel_stats.csv profiles/X.json <steps_profiled>"""
import csv
import json
import sys


def main(stats_csv, bench_json, steps):
    steps = int(steps)
    rows = list(csv.DictReader(open(stats_csv)))
    conv = [r for r in rows if "conv_halo_kernel" in r["Name"] or "conv_wide_kernel" in r["Name"] or "vol32_kernel" in r["Name"] or "vol32_fused_kernel" in r["Name"] or "t_mask_kernel" in r["Name"]]
    warp = [r for r in rows if "grid_sample_kernel" in r["Name"] or "dm_softmax_warp_kernel" in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
    ctot = sum(float(r["TotalDurationNs"]) for r in conv) / 1e6
    ccalls = sum(int(r["Calls"]) for r in conv)
    d = json.loads(open(bench_json).read().strip().splitlines()[-1])
    rf = d["roofline"]
    print(f"rocprof: {len(rows)} kernels, {tot / steps:.3f} ms of kernel time per step over {steps} steps")
    print(f"rocprof: convolution kernels {ctot / steps:.3f} ms/step, {ccalls // steps} launches/step, avg {ctot * 1e3 / ccalls:.2f} us/launch")
    print(f"bench  : conv {rf['conv_ms_per_step']} ms/step, {rf['launches_per_step']} launches/step, avg {rf['avg_launch_us']} us/launch "
          f"-> {rf['achieved']} TFLOP/s = {rf['frac']} of {rf['peak']}")
    if warp:
        w = warp[0]
        print(f"rocprof: grid_sample_kernel avg {float(w['AverageNs']) / 1e3:.2f} us/launch; bench: {d.get('warp_roofline', {}).get('avg_launch_us')} us")
    print("top kernels (ms/step):")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
        print(f"  {float(r['TotalDurationNs']) / 1e6 / steps:8.3f}  {int(r['Calls']) // steps:4d}x  {r['Name'][:100]}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
