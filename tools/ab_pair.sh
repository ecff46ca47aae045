#!/bin/bash
# Same-box A/B of the step: bench.py on the in-tree libudet.so ("new") alternating with bench.py on another build of the library ("base",
# through tools/ab_bench.py) inside ONE gpurun call -- boxes differ by up to 10 %, pairs on one box do not.  The base library is not in
# the repository: build it from the revision to compare with (make -C unsupervised_detection_amd/csrc at that revision) and copy it to
# gpurun_exp/libudet_base.so (git-ignored, but it travels to the GPU box).  Output: profiles/r06_ab_ladder.txt has the round's runs.
mkdir -p gpurun_out/r6t; O=gpurun_out/r6t
A="--steps 40 --no-cpu-baseline --cycles 0 --ensemble-frames 0"
for i in 1 2; do
python bench.py $A > $O/new$i.json 2>/dev/null
python tools/ab_bench.py --lib gpurun_exp/libudet_base.so -- $A > $O/base$i.json 2>/dev/null
done
python - <<'P'
import json
for n in ['new1','base1','new2','base2']:
    d=json.loads([l for l in open(f'gpurun_out/r6t/{n}.json') if l.startswith('{')][-1]); r=d['roofline']
    print(n, d['ms_per_step'], d['ms_per_step_median'], r['conv_ms_per_step_serial'], r['families']['direct']['ms_per_step_serial'], r['families']['winograd_f2x2_3x3']['ms_per_step_serial'], r['frac_algorithmic'])
P
