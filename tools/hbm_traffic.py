"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; `--pmc X --kernel-trace --output-format csv`) of
`bench.py --steps 3 --warmup 1 --sample-steps 0 --no-cpu-baseline` into profiles/rNN_hbm_traffic.json:
memory-side bytes per launch for every contraction-kernel class (what bench.py reports as roofline.traffic).

    python tools/hbm_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section): rocprofv3 prints both counters in KiB;
FETCH_SIZE reports half the bytes of wide (16 B/lane) coalesced reads, which is how every contraction kernel reads
(global_load_dwordx4 / global_load_lds_dwordx4) -> x2; WRITE_SIZE is exact for 16-B stores and float atomics.
Infinity-Cache hits are counted by both, so this is L2-miss (fabric) traffic, an upper bound on HBM bytes."""
import csv
import json
import re
import sys
from collections import defaultdict


def klass(name):
    m = re.match(r"(?:void )?(igemm|wgrad)(?:_glds)?_kernel<(\d+), (\d+)", name)
    if not m:
        return None
    if m.group(1) == "igemm":
        return f"igemm_{m.group(2)}x{m.group(3)}"
    return f"wgrad_{m.group(2)}x{m.group(2)}"


def fold(path, counter):
    per = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            e = per[row["Kernel_Name"]]
            e[0] += 1
            e[1] += float(row["Counter_Value"])
    return per


def main():
    fetch, write, out = sys.argv[1:4]
    fr, wr = fold(fetch, "FETCH_SIZE"), fold(write, "WRITE_SIZE")
    classes = defaultdict(lambda: dict(launches=0, read=0.0, write=0.0))
    raw = {}
    for name, (n, kib) in fr.items():
        wn, wkib = wr.get(name, (0, 0.0))
        raw[name[:140]] = dict(launches=n, FETCH_SIZE_KB_per_launch=kib / n, WRITE_SIZE_KB_per_launch=(wkib / wn if wn else None))
        c = klass(name)
        if c:
            classes[c]["launches"] += n
            classes[c]["read"] += kib * 1024 * 2
            classes[c]["write"] += (wkib / wn * n if wn else 0.0) * 1024
    res = dict(
        command="rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python bench.py --steps 3 --warmup 1 "
                "--sample-steps 0 --no-cpu-baseline  (second pass: --pmc WRITE_SIZE); folded by tools/hbm_traffic.py",
        units="rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB",
        correction="MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reads 1/2 of the bytes of wide (16 B/lane) coalesced reads "
                   "-> x2 for the contraction kernels (16-B global_load / global_load_lds); WRITE_SIZE is exact for 16-B stores "
                   "and float atomics; Infinity-Cache hits are included (L2-miss traffic)",
        per_class_hbm_bytes_per_launch={
            c: dict(launches=v["launches"], read_bytes_corrected=v["read"] / v["launches"], write_bytes=v["write"] / v["launches"],
                    total=(v["read"] + v["write"]) / v["launches"]) for c, v in sorted(classes.items())},
        per_kernel_raw=raw)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    for c, v in res["per_class_hbm_bytes_per_launch"].items():
        print(f"{c:16s} launches {v['launches']:5d}  read {v['read_bytes_corrected']/1e6:8.2f} MB  write {v['write_bytes']/1e6:8.2f} MB")


if __name__ == "__main__":
    main()
