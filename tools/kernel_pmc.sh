#!/bin/bash
# PMC counters of the kernels whose name contains $1, in one rocprofv3 pass per counter group (never combined with the system / HIP
# trace domains); prints the median of every counter per kernel name.
#   bash tools/kernel_pmc.sh conv_thin -- python tools/conv_bench.py --only rec.flow1 --reps 3
FILTER=$1; shift; [ "$1" = "--" ] && shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/kpmc_$i
  (cd "$R" && rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/kpmc_$i -o p -- "$@") > /tmp/kpmc_$i.log 2>&1
done
KPMC_FILTER="$FILTER" python3 - <<'PY'
import csv, glob, collections, os
flt = os.environ["KPMC_FILTER"]
vals = collections.defaultdict(dict)
for path in glob.glob('/tmp/kpmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if flt not in k: continue
        k = k[:110]
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        d = vals[k]
        d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        d.setdefault("us", []).append(dur)
        for f in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
            if f in r: d.setdefault(f, []).append(float(r[f]))
for k, d in sorted(vals.items()):
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print("   %-28s %.5g   (n=%d)" % (c, v[len(v) // 2], len(v)))
PY
