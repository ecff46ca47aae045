#!/bin/bash
# PMC counters of the Winograd prototype kernels (tools/wino_bench.hip built as gpurun_exp/wino_bench), one rocprofv3 pass per group.
#   bash tools/wino_pmc.sh dc_conv21 > gpurun_out/wino_pmc.txt
cd /tmp && export TMPDIR=/tmp
BIN=$GRAFT_REPO_ROOT/gpurun_exp/wino_bench
[ -x "$BIN" ] || BIN=/root/repo/gpurun_exp/wino_bench
i=0
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf /tmp/wpmc_$i
  rocprofv3 --kernel-trace --pmc $g --output-format csv -d /tmp/wpmc_$i -o p -- $BIN "$@" > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, collections
vals = collections.defaultdict(dict)
for path in glob.glob('/tmp/wpmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "wino" not in k or "weights" in k: continue
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        d = vals[k]
        d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        d.setdefault("us", []).append(dur)
for k, d in sorted(vals.items()):
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print("   %-28s %.4g" % (c, v[len(v) // 2]))
PY
