#!/bin/bash
# usage: tools/pmc_one.sh <lib-variant|default> <outdir> <counters...>   (runs tools/bench_kernels.py under rocprofv3 --pmc)
V=$1; OUT=$2; shift 2
export TMPDIR=/tmp
mkdir -p $OUT
if [ "$V" != "default" ]; then export UFR_LIB=$PWD/uforecon_amd/lib/libufr_$V.so; fi
timeout 420 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT -o p -- python3 tools/bench_kernels.py > /dev/null 2> $OUT/err.txt
python3 - "$OUT" <<PY
import csv, collections, sys
acc=collections.defaultdict(lambda: collections.defaultdict(list)); dur=collections.defaultdict(list); seen=set()
for r in csv.DictReader(open(sys.argv[1]+"/p_counter_collection.csv")):
    k=r["Kernel_Name"].split("(")[0][-40:]
    if "transformer" not in k and "gather" not in k: continue
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); dur[k].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
for k,v in acc.items():
    print(k, "avg_us %.1f" % (sum(dur[k])/len(dur[k])/1e3), {c: "%.4g" % (sum(x)/len(x)) for c,x in v.items()})
PY
