python -m pytest tests/test_grad.py tests/test_trainer_fused.py tests/test_grad_stages.py -x -q 2>&1 | tail -1
for i in 1 2 3; do for f in 1 0; do echo "fold=$f"; OARD_GATE_FOLD=$f python bench.py --mode train --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); t=d[\"train_step\"]; print(round(d[\"ms_per_step\"],2), t[\"families_ms_per_step\"])"; done; done
