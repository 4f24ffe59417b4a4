#!/bin/bash
# A/B of the SYRK tile order: HBM-side fetch traffic (PMC) and duration with / without XCD-private super-tiles
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/supertile
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ST in 0 1; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$ST -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --supertile $ST > $OUT/fetch_$ST.log 2>&1
  timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc_$ST -o bench -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --supertile $ST > $OUT/tcc_$ST.log 2>&1
  timeout 300 python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras --supertile $ST > $OUT/bench_$ST.json 2> $OUT/bench_$ST.err
done
python3 - <<PY
import csv, collections, json
for st in (0, 1):
    for sub in ("fetch", "tcc"):
        agg = collections.defaultdict(list)
        try:
            with open("$OUT/%s_%d/bench_counter_collection.csv" % (sub, st)) as f:
                for row in csv.DictReader(f):
                    if "gemm_nt_kernel<double, 0>" in row["Kernel_Name"]:
                        agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        except Exception as e:
            print("missing", sub, st, e)
        for c, v in agg.items():
            print("supertile=%d %s avg/launch %.5g (n=%d)" % (st, c, sum(v) / len(v), len(v)))
    try:
        b = json.load(open("$OUT/bench_%d.json" % st))
        print("supertile=%d ms_per_step %.2f syrk TFLOP/s %.2f" % (st, b["ms_per_step"], b["roofline"]["achieved"]))
    except Exception as e:
        print("bench missing", st, e)
PY
find $OUT -name "*.csv" -size +2M -delete
