"""Stage-by-stage check of the training path on the GPU box (prints; tests/test_grad_stages.py asserts the same numbers).
usage: python tools/grad_debug.py [case]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
from _stage_checks import run  # noqa: E402

if __name__ == "__main__":
    out, errs, flat, gap = run(sys.argv[1] if len(sys.argv) > 1 else "g9_grad_h32")
    print(f"whole step: flat gradient error {flat:.2e}")
    for n in sorted(errs, key=lambda k: -errs[k]):
        print(f"  {n:62s} ours {errs[n]:.2e}   reference f32 {gap[n]:.2e}")
