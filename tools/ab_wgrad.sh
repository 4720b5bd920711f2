python -m pytest tests/test_wgrad.py -q 2>&1 | tail -5
for m in 0 256 512; do echo "wgrad_lds=$m"; OARD_WGRAD_LDS=$m python bench.py --mode train --steps 6 --warmup 2 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['train_step']['families_ms_per_step'], d['train_step']['tflops_by_family'])"; done
