#!/bin/bash
# Re-collects every measurement artefact under profiles/ on an MI355X box (one GPU):
#   gpurun --timeout 1800 -- 'bash tools/refresh_profiles.sh'        then copy gpurun_out/prof/* to profiles/rNN_*
# 1. the bench line (tuning run: writes the configurations it picked and the per-launch CSV of its measurement pass)
# 2. rocprofv3 kernel traces of TIMED STEPS ONLY (serial and pipelined), reduced by tools/trace_report.py
# 3. PMC counters of whole steps in separate passes (tools/pmc_step.py; never combined with the system trace domains)
# 4. the HBM-bound warp / cost-volume kernels and the MFMA ceiling probe
# 5. the fp16 mode, fp32 at batch 2, the one-rank RCCL run
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof
mkdir -p "$O"
cd "$R"
UDET_TUNE_LOG=1 UDET_PROF_DUMP=$O/layers.csv python bench.py --tune-cache "$O/tune.txt" > "$O/bench.json" 2> "$O/bench.err"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_s /tmp/tr_p
UDET_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_s -o steps -- \
    python "$R/bench.py" --tune-cache "$O/tune.txt" --trace-only --no-pipeline --steps 10 --warmup 2 > "$O/trace_serial.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_p -o steps -- \
    python "$R/bench.py" --tune-cache "$O/tune.txt" --trace-only --steps 10 --warmup 2 > "$O/trace_pipelined.log" 2>&1
cd "$R"
python tools/trace_report.py /tmp/tr_s --steps 12 --layers "$O/layers.csv" --out "$O/trace_report_serial.json" \
    --copy-stats "$O/kernel_stats_serial_steps.csv" > /dev/null
python tools/trace_report.py /tmp/tr_p --steps 12 --layers "$O/layers.csv" --out "$O/trace_report_pipelined.json" \
    --copy-stats "$O/kernel_stats_pipelined_steps.csv" > /dev/null
cd /tmp
python "$R/tools/pmc_step.py" --tune-cache "$O/tune.txt" --out "$O/pmc_step.json" > /dev/null 2> "$O/pmc_step.err"
cd "$R"
python tools/cv_bench.py > "$O/cv_bench.txt" 2>&1
# 5. BASELINE configs[4] (fp16 convolution GEMMs, batch 2) beside fp32 at that batch; the RCCL branch at world size 1 (allreduce_ms of a one-rank group)
python bench.py --fp16-convs --cycles 0 --ensemble-frames 0 > "$O/bench_fp16_convs.json" 2> "$O/bench_fp16.err"
python bench.py --batch 2 --cycles 0 --ensemble-frames 0 --no-cpu-baseline > "$O/bench_fp32_batch2.json" 2> /dev/null
UDET_DP_WORLD1=1 python bench.py --tune-cache "$O/tune.txt" --pmc-json "$O/pmc_step.json" --cycles 0 --ensemble-frames 0 --no-cpu-baseline > "$O/bench_rccl_world1.json" 2> "$O/bench_rccl_world1.err"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mfma_peak tools/mfma_peak.hip 2> /dev/null && /tmp/mfma_peak > "$O/mfma_peak.txt" 2>&1
# 6. round 6: the F(4x4,3x3) prototype with its ablations, the fixed-cost anatomy of the Winograd / LDS-DMA launches (libudet_exp.so:
#    make -C unsupervised_detection_amd/csrc exp), the Winograd per-stage / intercept fit
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o /tmp/wino43_bench tools/wino43_bench.hip 2> /dev/null && W43_TS=1 W43_ABL=1 /tmp/wino43_bench dc_conv21 conv2_1 conv2_3 dc_conv31 gen.conv5 > "$O/wino43_proto.txt" 2>&1
if [ -f unsupervised_detection_amd/libudet_exp.so ]; then
  python tools/wino_stamps.py 2> /dev/null | grep -v amdgpu.ids > "$O/launch_anatomy.txt"
  python tools/igemm_stamps.py 2> /dev/null | grep -v amdgpu.ids | grep -v "d=8" >> "$O/launch_anatomy.txt"
fi
python tools/wino_fixed_cost.py 2> /dev/null | grep -v amdgpu.ids > "$O/wino_fixed_cost.txt"
# 7. the bench line once more, on the configurations of step 1 and with THIS call's counters in roofline.traffic (step 1 ran before them)
mv "$O/bench.json" "$O/bench_tuning_run.json"
python bench.py --tune-cache "$O/tune.txt" --pmc-json "$O/pmc_step.json" > "$O/bench.json" 2> "$O/bench_final.err"
ls -la "$O"
