#!/bin/bash
# per-kernel prefill times: tools/prefill_prof.sh <tag> [BxP ...]   -> gpurun_out/pfprof_<tag>_<BxP>.csv
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=$1; shift
for pf in "${@:-32x128}"; do
  rm -rf /tmp/pfprof && PF=$pf rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfprof -o p -- python3 tools/prefill_prof.py > /tmp/pfprof.log 2>&1 || { tail -5 /tmp/pfprof.log; exit 1; }
  tail -1 /tmp/pfprof.log
  f=$(find /tmp/pfprof -name "*kernel_stats.csv" | head -1)
  mkdir -p gpurun_out && cp "$f" gpurun_out/pfprof_${tag}_${pf}.csv
  echo "== $tag $pf"; head -12 "$f" | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-150
done
