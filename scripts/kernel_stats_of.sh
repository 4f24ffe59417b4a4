# bash scripts/kernel_stats_of.sh <what> <N> [dtype]   -> top kernels of that API call (see gpu_kernel_stats_of.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kso
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kso -o k -- python3 $R/scripts/gpu_kernel_stats_of.py "$@" > /dev/null 2>&1
python3 - "$@" <<'EOF2'
import csv, glob, sys
f = glob.glob("/tmp/kso/**/k_kernel_stats.csv", recursive=True)[0]
print("##", " ".join(sys.argv[1:]), "(4 calls + one fit)")
for r in list(csv.DictReader(open(f)))[:10]:
    print(f"  {r['Name'].split('(')[0].replace('void gphip::', '')[:70]:70s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:9.1f} us  total {float(r['TotalDurationNs']) / 1e6:8.2f} ms")
EOF2
