"""Condense a rocprofv3 output directory (gpurun_out/prof_rN) into the small tracked files under
profiles/: kernel stats (names shortened) and a per-kernel PMC table.

    python tools/summarize_prof.py gpurun_out/prof_r1 profiles/r1
"""
import collections
import csv
import json
import os
import sys


def short(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "").split("(")[0].strip()
    for pre in ("void ", "ufr::"):
        name = name.replace(pre, "")
    return name[-60:]


def pmc_tables(src: str, dst_prefix: str, group: str):
    table = collections.defaultdict(dict)
    for base in ("pmc_sq", "pmc_fetch", "pmc_write", "pmc_lds", "pmc_tcc"):
        tag = group + base
        p = os.path.join(src, tag + "_counter_collection.csv")
        if not os.path.exists(p):
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(list)
        seen = set()
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for k, v in acc.items():
            for c, x in v.items():
                table[k][c] = sum(x) / len(x)
            table[k].setdefault("launches", len(next(iter(v.values()))))
            table[k][f"avg_ns[{base}]"] = sum(dur[k]) / max(len(dur[k]), 1)
    if not table:
        return
    with open(dst_prefix + "_pmc_summary.md", "w") as g:
        g.write("# rocprofv3 PMC summary (mean per launch; separate --pmc passes)\n\n")
        g.write("FETCH_SIZE / WRITE_SIZE are in KiB as reported; on gfx950 FETCH_SIZE counts half of the bytes of wide\n"
                "coalesced reads (MI355X_MICROARCH.md, HBM section): double it before comparing with byte counts.\n"
                "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES are cycles summed over SIMDs.\n\n")
        for k in sorted(table, key=lambda k: -table[k].get("SQ_WAVE_CYCLES", table[k].get("FETCH_SIZE", 0))):
            if "kernel" not in k:
                continue
            g.write(f"## {k}\n\n| counter | mean per launch |\n|---|---|\n")
            for c, v in sorted(table[k].items()):
                g.write(f"| {c} | {v:,.1f} |\n")
            t = table[k]
            if "SQ_VALU_MFMA_BUSY_CYCLES" in t and t.get("avg_ns[pmc_sq]") and t["SQ_VALU_MFMA_BUSY_CYCLES"] > 0:
                per_simd = t["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
                g.write(f"\nMFMA-busy cycles per SIMD: {per_simd:,.0f} over {t['avg_ns[pmc_sq]'] / 1e3:,.1f} us "
                        f"=> {per_simd / (t['avg_ns[pmc_sq]'] * 2.4):.2%} of a 2.4 GHz clock\n")
            g.write("\n")
    # HBM bytes per launch, corrected as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE (KiB) counts
    # half the bytes of wide coalesced reads on gfx950 -> x2; WRITE_SIZE (KiB) taken as reported (uncalibrated)
    for k, t in table.items():
        if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
            t["hbm_bytes_per_launch"] = (2.0 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024.0
    with open(dst_prefix + "_pmc.json", "w") as g:
        json.dump({k: v for k, v in table.items() if "kernel" in k}, g, indent=1)
        g.write("\n")


def main(src: str, dst_prefix: str):
    os.makedirs(os.path.dirname(dst_prefix), exist_ok=True)
    stats = os.path.join(src, "stats_kernel_stats.csv")
    if os.path.exists(stats):
        with open(stats) as f, open(dst_prefix + "_kernel_stats.csv", "w", newline="") as g:
            w = csv.writer(g)
            for i, row in enumerate(csv.reader(f)):
                if i:
                    row[0] = short(row[0])
                w.writerow(row)
    # training step (tools/bench_train.py), fp32 / 16-bit matrix mode / with cost_reg_2 in front; the per-frame producers
    for tag in ("train", "train16", "traincr", "encoder"):
        tstats = os.path.join(src, f"{tag}_kernel_stats.csv")
        if os.path.exists(tstats):
            with open(tstats) as f, open(f"{dst_prefix}_{tag}_kernel_stats.csv", "w", newline="") as g:
                w = csv.writer(g)
                for i, row in enumerate(csv.reader(f)):
                    if i:
                        row[0] = short(row[0])
                    w.writerow(row)
        tl = os.path.join(src, f"{tag}_line.json")
        if os.path.exists(tl):
            for line in open(tl).read().strip().splitlines():
                if line.startswith("{"):
                    with open(f"{dst_prefix}_{tag}_under_rocprof.json", "w") as g:
                        json.dump(json.loads(line), g, indent=1)
                        g.write("\n")
    # counter passes: "" = bench.py configs[1], "train_" = tools/bench_train.py, "c4_" = tools/bench_c4.py (configs[3])
    for group in ("", "train_", "c4_"):
        pmc_tables(src, dst_prefix + ("_" + group.rstrip("_") if group else ""), group)
    bl = os.path.join(src, "bench_line.json")
    if os.path.exists(bl):
        txt = open(bl).read().strip().splitlines()
        for line in txt:
            if line.startswith("{"):
                with open(dst_prefix + "_bench_under_rocprof.json", "w") as g:
                    json.dump(json.loads(line), g, indent=1)
                    g.write("\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
