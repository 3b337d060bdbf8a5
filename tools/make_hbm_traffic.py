"""profiles/hbm_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/round_profile.sh:
    python tools/make_hbm_traffic.py gpurun_out/r02_d 32 > profiles/hbm_traffic.json
Per convolution launch (conv_halo / conv_igemm kernels): FETCH_SIZE x 2 (gfx950 tallies the 128-byte requests of wide reads at
64 bytes, MI355X_MICROARCH.md) + WRITE_SIZE (uncalibrated); rocprofv3 reports both in KB."""
import collections
import csv
import json
import sys


def per_launch(path, counter):
    tot, seen = 0.0, set()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        if "conv_halo_kernel" not in r["Kernel_Name"] and "conv_igemm_kernel" not in r["Kernel_Name"]:
            continue
        tot += float(r["Counter_Value"])
        seen.add(r["Dispatch_Id"])
    return tot * 1024.0 / max(len(seen), 1), len(seen)


root, batch = sys.argv[1], int(sys.argv[2])
f, nf = per_launch(f"{root}/pmc_FETCH_SIZE/pmc_counter_collection.csv", "FETCH_SIZE")
w, nw = per_launch(f"{root}/pmc_WRITE_SIZE/pmc_counter_collection.csv", "WRITE_SIZE")
print(json.dumps({"batch": batch, "conv_launches_counted": nf, "fetch_bytes_per_conv_launch_x2": int(2 * f),
                  "write_bytes_per_conv_launch_raw": int(w),
                  "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python bench.py --steps 2 --warmup 1; FETCH_SIZE "
                            "doubled per MI355X_MICROARCH.md (gfx950 tallies 128-byte requests at 64 bytes), WRITE_SIZE uncalibrated"}, indent=1))
