# which leg of the default bench.py run changes the train_step leg (none does: 70.5-71.1 ms in every combination; one default run on
# one box measured 76.9 ms, a repeat 70.8): usage tools/bench_legs_probe.sh
for sk in "second_line,config5,sampler" "second_line,config5,roofline" "second_line,config5,roofline,sampler" "second_line,config5"; do
OARD_BENCH_SKIP=$sk python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d['train_step']; print('skip $sk', 'train ms', round(t['ms_per_step'],2), 'kernels', t['hip_kernel_ms_per_step'])"
done
