#!/bin/bash
# GPU box: FETCH_SIZE (x2: gfx950 tallies 128-byte requests at 64) and duration per conv_wide instantiation for library builds:  bash tools/ab_fetch.sh TAG "" ab/x.so
TAG=$1; shift
O=/root/repo/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  n=$(basename "${v:-base}" .so)
  CANONSWAP_LIB=${v:+/root/repo/$v} rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_$n -o pmc --output-format csv -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fixed-job > $O/pmc_$n.json 2> $O/pmc_$n.err
  python - <<PY | tee -a $O/fetch.txt
import csv, glob, collections
f = glob.glob("$O/pmc_$n/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] != "FETCH_SIZE": continue
    k = r["Kernel_Name"]
    if "conv_wide" not in k: continue
    a = acc[k]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, (n_, v_, t_) in sorted(acc.items()):
    print("$n", k[-40:], "launches", n_, "FETCH x2 per launch %.3f GB" % (2 * v_ * 1024 / n_ / 1e9 if v_ / n_ < 1e7 else 2 * v_ / n_ / 1e9), "us %.1f" % (t_ / n_ / 1e3))
PY
  rm -rf $O/pmc_$n
done
