#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as the MI355X guide prescribes)
into HBM bytes per launch of the LZ4 decode path.

Corrections applied (see /opt/skills/guides/MI355X_MICROARCH.md §HBM):
  * counters are in KiB;
  * on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced (16 B/lane) streaming reads, which is
    what both kernels issue for the compressed stream -> reads are doubled;
  * WRITE_SIZE is used as is: it was calibrated here on a known byte count — the LDS decoder writes exactly
    n_chunks x 65536 B with 16 B/lane stores and WRITE_SIZE x 1024 reproduces that number to the byte.
usage: tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def main():
    fetch, nf = per_kernel(sys.argv[1])
    write, _ = per_kernel(sys.argv[2])
    out = {"unit": "bytes per launch",
           "corrections": "KiB->B; FETCH_SIZE x2 (gfx950 wide coalesced reads; uncalibrated upper bound for the lane kernel's divergent 16 B reads); "
                          "WRITE_SIZE as is (calibrated on the LDS decoder's exact output size)",
           "kernels": {}}
    total = 0.0
    for k in sorted(fetch):
        if "lz4" not in k:
            continue
        rd = fetch[k] * 1024 * 2
        wr = write.get(k, 0.0) * 1024
        out["kernels"][k] = {"launches_sampled": nf[k], "hbm_read_bytes": rd, "hbm_write_bytes": wr}
        total += rd + wr
    out["total_hbm_bytes_per_step"] = total
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
