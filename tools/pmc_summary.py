"""Summarise the counter CSVs written by tools/pmc_passes.sh: per-kernel sums over dispatches.

    python3 tools/pmc_summary.py gpurun_out/pmc_<tag> [kernel-substring] > profiles/<name>.md
"""
import csv, glob, os, sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else "pt_render"
    tot = defaultdict(float)
    ndisp = defaultdict(set)
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for row in csv.DictReader(open(f)):
            if sub not in row["Kernel_Name"]:
                continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            ndisp[row["Counter_Name"]].add((f, row["Dispatch_Id"]))
    print("# PMC summary for kernels matching %r in %s" % (sub, d))
    print("| counter | per-dispatch value | dispatches |")
    print("|---|---|---|")
    per = {}
    for k in sorted(tot):
        n = len(ndisp[k])
        per[k] = tot[k] / n
        print("| %s | %.6g | %d |" % (k, per[k], n))
    g = per.get
    print()
    if g("SQ_WAVE_CYCLES") and g("SQ_ACTIVE_INST_VALU"):
        print("- VALU-active share of wave cycles: %.3f" % (g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")))
    if g("SQ_WAVE_CYCLES") and g("SQ_WAIT_ANY"):
        print("- waiting (s_waitcnt/barrier) share of wave cycles: %.3f" % (g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES")))
    if g("SQ_INSTS_VALU") and g("SQ_WAVES"):
        print("- VALU instructions per wave: %.1f ; SALU %.1f ; SMEM %.1f" %
              (g("SQ_INSTS_VALU") / g("SQ_WAVES"), g("SQ_INSTS_SALU", 0) / g("SQ_WAVES"), g("SQ_INSTS_SMEM", 0) / g("SQ_WAVES")))
    if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
        print("- mean active lanes per VALU instruction (divergence): %.1f of 64" %
              (g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU")))
    if g("GRBM_GUI_ACTIVE"):
        print("- GRBM_GUI_ACTIVE / 8 XCDs = %.4g shader-clock cycles per dispatch" % (g("GRBM_GUI_ACTIVE") / 8))
    if g("FETCH_SIZE") is not None:
        print("- FETCH_SIZE %.4g KB per dispatch (x2 for wide coalesced reads on gfx950, MI355X_MICROARCH.md HBM section)" % g("FETCH_SIZE"))
    if g("WRITE_SIZE") is not None:
        print("- WRITE_SIZE %.4g KB per dispatch" % g("WRITE_SIZE"))


if __name__ == "__main__":
    main()
