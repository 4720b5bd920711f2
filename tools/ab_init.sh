# A/B of the init-stage backward kernels inside the training step: parity tests, then the family times of bench.py --mode train
python -m pytest tests/test_grad.py tests/test_grad_stages.py tests/test_trainer_fused.py -x -q 2>&1 | tail -1
for i in 1 2; do python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d[\"train_step\"]; print(round(d[\"ms_per_step\"],2), t[\"families_ms_per_step\"])"; done
