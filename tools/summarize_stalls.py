"""Per-kernel wave-state / memory-pipe summary of the PMC passes written by tools/pmc_stalls.sh.
    python tools/summarize_stalls.py gpurun_out/stalls > profiles/rNN_wave_state.txt
wait_any = wave parked on s_waitcnt / barrier, wait_inst = issue stall (pipe busy / dependency), active = issuing;
the three are fractions of SQ_WAVE_CYCLES.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)."""
import collections
import csv
import glob
import sys


def main(root):
    val = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    dur = collections.defaultdict(float)
    nd = collections.Counter()
    for i, f in enumerate(sorted(glob.glob(f"{root}/p*/pmc_counter_collection.csv"))):
        seen = set()
        for r in csv.DictReader(open(f)):
            k, c = r["Kernel_Name"], r["Counter_Name"]
            val[k][c] += float(r["Counter_Value"]); n[k][c] += 1
            if i == 0 and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); nd[k] += 1
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])

    def g(k, c):
        return val[k][c] / max(n[k][c], 1)

    print(f"{'kernel':70s} {'n':>4s} {'us':>7s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>6s} {'mfma_busy':>9s} {'valu/mfma':>9s} "
          f"{'vmem_rd/mfma':>12s} {'lds/mfma':>8s} {'lds_conf':>8s} {'L1acc/L2req':>11s}")
    for k in sorted(val, key=lambda k: -dur[k])[:24]:
        wc = g(k, "SQ_WAVE_CYCLES") or 1
        gui = g(k, "GRBM_GUI_ACTIVE") / 8 or 1
        mf = g(k, "SQ_INSTS_MFMA")
        la = g(k, "SQ_LDS_IDX_ACTIVE")
        print(f"{k[:70]:70s} {nd[k]:4d} {dur[k] / max(nd[k], 1) / 1e3:7.0f} {g(k, 'SQ_WAIT_ANY') / wc:8.2f} {g(k, 'SQ_WAIT_INST_ANY') / wc:9.2f} "
              f"{g(k, 'SQ_ACTIVE_INST_ANY') / wc:6.2f} {g(k, 'SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / gui:9.2f} "
              f"{(g(k, 'SQ_INSTS_VALU') / mf if mf else 0):9.2f} {(g(k, 'SQ_INSTS_VMEM_RD') / mf if mf else 0):12.3f} "
              f"{(g(k, 'SQ_INSTS_LDS') / mf if mf else 0):8.3f} {(g(k, 'SQ_LDS_BANK_CONFLICT') / la if la else 0):8.2f} "
              f"{g(k, 'TCP_TOTAL_CACHE_ACCESSES') / max(g(k, 'TCP_TCC_READ_REQ'), 1):11.2f}")


if __name__ == "__main__":
    main(sys.argv[1])
