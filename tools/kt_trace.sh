#!/bin/bash
# tools/kt_trace.sh <tag> <python script + args> — rocprofv3 kernel trace of a script on the GPU box: prints the kernel stats and
# the duration of each render / walk / fold kernel of the last launches, in start order.  Output under gpurun_out/<tag>/.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
(cd $R && python3 -c 'from chunkyclplugin_amd import native; native.build(); native.lib(); from oracle import binding; binding.port()') || exit 1
export CHUNKY_ORACLE_NO_BUILD=1   # nothing is compiled or spawned from inside the profiled process
cd /tmp && export TMPDIR=/tmp
(cd $R && timeout ${KT_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 "$@" > $O/kt.log 2>&1 < /dev/null)
cd $R
f=$(find $O/kt -name "*kernel_stats.csv" 2>/dev/null | head -1)
t=$(find $O/kt -name "*kernel_trace.csv" 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats.csv && cut -c1-160 $O/kernel_stats.csv | head -8
[ -n "$t" ] && python3 - "$t" <<'PY' | tee $O/sequence.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("render_", "walk_kernel", "fold_kernel"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows[-24:]:
    print("%-60s %9.3f ms" % (r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
PY
rm -rf $O/kt
tail -2 $O/kt.log | cut -c1-300
