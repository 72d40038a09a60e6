set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in "solow 4096" "trade 8192"; do
  set -- $k
  rm -rf gpurun_out/fu_$1
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fu_$1 -- python3 tools/flat_update_prof.py $1 $2 > gpurun_out/fu_$1.log 2>&1
  python3 - $1 <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/fu_%s/*/*_kernel_stats.csv" % sys.argv[1])[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:9]:
    print("%-70s calls %5s avg %9.1f us %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
  rm -rf gpurun_out/fu_$1
done
