"""Fold rocprofv3 PMC passes of ONE kernel into a JSON summary (profiles/rNN_<kernel>_pmc.json).

    python tools/pmc_fold.py <kernel-name substring> <out.json> "<shape description>" <flops> <bytes> <pass_dir> [<pass_dir> ...]

Each pass_dir is the -d directory of one `rocprofv3 --pmc <group> --kernel-trace --output-format csv` run (one counter
group per pass, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE cannot share a pass).  The LAST dispatch
of the kernel in every pass is taken (warm caches).  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950
FETCH_SIZE counts half the bytes of 16-B-per-lane reads (global_load_dwordx4 / global_load_lds_dwordx4) -> x2."""
import csv
import glob
import json
import sys


def main():
    sub, out, shape, flops, nbytes = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), float(sys.argv[5])
    counters, durs, kname, meta = {}, [], None, {}
    for d in sys.argv[6:]:
        for f in glob.glob(d + "/*/*counter_collection.csv"):
            last = {}
            for row in csv.DictReader(open(f)):
                if sub in row["Kernel_Name"]:
                    kname = row["Kernel_Name"]
                    last.setdefault(row["Dispatch_Id"], {})[row["Counter_Name"]] = float(row["Counter_Value"])
                    meta = dict(vgpr=row.get("VGPR_Count"), lds=row.get("LDS_Block_Size"), grid=row.get("Grid_Size"))
            if last:
                counters.update(last[sorted(last, key=int)[-1]])
        for f in glob.glob(d + "/*/*kernel_trace.csv"):
            rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
            if rows:
                durs.append(int(rows[-1]["End_Timestamp"]) - int(rows[-1]["Start_Timestamp"]))
    ns = sum(durs) / max(1, len(durs))
    c = counters
    der = {"duration_ns_mean_over_passes": ns, "tflops": flops / ns / 1e3 if ns else None,
           "algorithmic_GBps": nbytes / ns if ns else None}
    if "SQ_WAVE_CYCLES" in c:
        tot = c["SQ_WAIT_INST_ANY"] + c["SQ_WAIT_ANY"] + c["SQ_ACTIVE_INST_ANY"]
        der["wave_time_split"] = {"issue_stalled (SQ_WAIT_INST_ANY)": c["SQ_WAIT_INST_ANY"] / tot,
                                  "waitcnt_or_barrier (SQ_WAIT_ANY)": c["SQ_WAIT_ANY"] / tot,
                                  "issuing (SQ_ACTIVE_INST_ANY)": c["SQ_ACTIVE_INST_ANY"] / tot}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # counts cycles summed over the 1024 SIMDs; a kernel of ns nanoseconds offers 1024 * ns * f_GHz SIMD-cycles
            der["mfma_busy_fraction_at_2.4GHz"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * ns * 2.4)
        if c.get("SQ_INSTS_MFMA"):
            der["valu_per_mfma"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"]
    if "SQ_LDS_BANK_CONFLICT" in c:
        der["lds_bank_conflict_fraction_of_lds_cycles"] = c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 1.0))
    if "FETCH_SIZE" in c:
        der["hbm_read_bytes_corrected_x2"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c:
        der["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        der["traffic_over_algorithmic"] = (c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024) / nbytes
    res = dict(kernel=kname, shape=shape, algorithmic_flops=flops, algorithmic_bytes=nbytes, counters_last_dispatch=c,
               duration_ns_per_pass=durs, derived=der, **meta)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(der, indent=1))


if __name__ == "__main__":
    main()
