"""Fold the three rocprofv3 PMC passes of tools/pmc_attn.sh (SQ group / FETCH_SIZE / WRITE_SIZE, one counter group per
pass) into profiles/rNN_attention_bf16_pmc.json: the kernels of the LAST attention-core call, in launch order.

    python tools/pmc_attn_fold.py <out.json> <n> <S> <d> <kernels per call> <pass_dir1> <pass_dir2> <pass_dir3>"""
import csv
import glob
import json
import sys

KEYS = ("igemm_bf16_kernel", "igemm8_bf16_kernel", "colstats_combine", "transpose_bf16", "col_softmax_bf16", "attn_")


def last_call(d, per_call):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    disp = {}
    for row in csv.DictReader(open(f)):
        if any(k in row["Kernel_Name"] for k in KEYS):
            e = disp.setdefault(int(row["Dispatch_Id"]), {"kernel": row["Kernel_Name"][:64]})
            e[row["Counter_Name"]] = float(row["Counter_Value"])
    ids = sorted(disp)[-per_call:]
    t = glob.glob(d + "/*/*kernel_trace.csv")[0]
    dur = {int(r["Dispatch_Id"]): int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(t))}
    return [dict(disp[i], duration_ns=dur.get(i)) for i in ids]


def main():
    out, n, S, dk, per_call = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    sq, fe, wr = (last_call(d, per_call) for d in sys.argv[6:9])
    kernels, tot_ns, busy, bytes_tot = [], 0, 0.0, 0.0
    for a, b, c in zip(sq, fe, wr):
        ns = a["duration_ns"]
        k = {"kernel": a["kernel"], "duration_ns": ns,
             "mfma_busy_fraction_at_2.4GHz": round(a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * ns * 2.4), 3),
             "hbm_read_bytes_corrected_x2": b.get("FETCH_SIZE", 0.0) * 2048, "hbm_write_bytes": c.get("WRITE_SIZE", 0.0) * 1024}
        if a.get("SQ_INSTS_MFMA"):
            k["valu_per_mfma"] = round(a["SQ_INSTS_VALU"] / a["SQ_INSTS_MFMA"], 2)
        tot = a.get("SQ_WAIT_INST_ANY", 0) + a.get("SQ_WAIT_ANY", 0) + a.get("SQ_ACTIVE_INST_ANY", 0)
        if tot:
            k["wave_time_split"] = {"issue_stalled": round(a["SQ_WAIT_INST_ANY"] / tot, 3), "waitcnt_or_barrier": round(a["SQ_WAIT_ANY"] / tot, 3),
                                    "issuing": round(a["SQ_ACTIVE_INST_ANY"] / tot, 3)}
        kernels.append(k)
        tot_ns += ns
        busy += a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        bytes_tot += k["hbm_read_bytes_corrected_x2"] + k["hbm_write_bytes"]
    flops = 4.0 * n * S * S * dk
    qkvo = 4.0 * n * S * dk * 2
    res = {"shape": f"attention core forward, bf16, {n} images, S={S} tokens, one head of d={dk}; last call, 3 rocprofv3 --pmc passes "
                    "(SQ group / FETCH_SIZE / WRITE_SIZE), tools/pmc_attn.sh",
           "kernels": kernels, "total_ns": tot_ns, "algorithmic_flops": flops, "tflops": round(flops / tot_ns / 1e3, 1),
           "mfma_busy_fraction_whole_core": round(busy / (1024 * tot_ns * 2.4), 3), "north_star_target_mfma": 0.4,
           "algorithmic_bytes_q_k_v_o": qkvo, "hbm_bytes_total": bytes_tot, "traffic_over_algorithmic": round(bytes_tot / qkvo, 2)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "kernels"}, indent=1))
    for k in kernels:
        print(k["kernel"][:50], k["duration_ns"], k["mfma_busy_fraction_at_2.4GHz"], int(k["hbm_read_bytes_corrected_x2"]), int(k["hbm_write_bytes"]))


if __name__ == "__main__":
    main()
