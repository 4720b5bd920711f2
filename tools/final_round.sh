#!/bin/bash
# everything the end of a round re-measures on the final sources: GPU suites (fp32 / split precision), the profile passes, the training
# timeline, the default bench line and the training lines.   usage: tools/final_round.sh <tag>     (outputs in gpurun_out/)
tag=${1:-round6}
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/${tag}_gpu_tests.txt; tail -1 gpurun_out/${tag}_gpu_tests.txt
OARD_GCL_B3=1 OARD_EQUI_B3=1 OARD_TRAIN_B3=1 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/${tag}_gpu_tests_split_precision.txt
tail -1 gpurun_out/${tag}_gpu_tests_split_precision.txt
bash tools/profile.sh $tag > gpurun_out/profile_$tag.log 2>&1
bash tools/profile_b3.sh $tag > gpurun_out/profile_b3_$tag.log 2>&1
bash tools/profile_cfg5.sh $tag > gpurun_out/profile_cfg5_$tag.log 2>&1
bash tools/profile_wgrad.sh ${tag}_wgrad > gpurun_out/profile_wgrad_$tag.log 2>&1
bash tools/timeline_train.sh $tag > /dev/null 2>&1
cat gpurun_out/${tag}_source_stamp.txt
# bench.py takes roofline.traffic from profiles/<tag>_*: the passes above ARE this round's (same sources, same box) - put them where it looks
cp gpurun_out/${tag}_source_stamp.txt gpurun_out/${tag}*_pmc_*.txt profiles/
python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/bench_err.log
python bench.py --mode train --steps 20 --warmup 3 > gpurun_out/${tag}_train_bench_line.json 2>/dev/null
python bench.py --mode train --batch 14 --steps 20 --warmup 3 > gpurun_out/${tag}_train_b14.json 2>/dev/null
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_bench_line.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"])
t = d["train_step"]; print(t["ms_per_step"], t["tflops_by_family"])
s = d["second_line"]; print(s["value"], s["roofline"]["traffic"])
print({k: (v["ms_per_step"], v["roofline"]["traffic"]) for k, v in d["config5"]["variants"].items()})
for f in ("${tag}_train_bench_line", "${tag}_train_b14"):
    x = json.load(open("gpurun_out/" + f + ".json")); print(f, x["ms_per_step"], x["value"])
PY
