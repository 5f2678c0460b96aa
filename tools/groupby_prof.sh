#!/bin/bash
# per-kernel times of one high-cardinality GROUP BY bench run.  usage: tools/groupby_prof.sh <outdir> <groups> <dist> [env...]
out=$1; g=$2; dist=$3; shift 3
mkdir -p $out; export TMPDIR=/tmp
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof_${g}_${dist} -o t --output-format csv -- python bench.py --workload c4 --groups $g --group-dist $dist --steps 10 --warmup 2 --no-extras --no-pmc --no-cpu-baseline > $out/prof_${g}_${dist}.log 2>&1
f=$(find $out/prof_${g}_${dist} -name "*kernel_stats.csv" | head -1)
echo "## $g $dist $@"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in [r for r in rows if "exon::" in r["Name"] or "Memset" in r["Name"] or "fill" in r["Name"].lower()][:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
