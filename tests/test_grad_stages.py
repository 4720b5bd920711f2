"""Teacher-forced stage tests of the training path: every torch stage function against the taped output of
the HIP kernel it restates, and every HIP backward kernel of the edge stages against torch autograd on the same taped inputs
with random cotangents (tests/_stage_checks.py).  Tolerance 1e-5 of the largest entry, as for the forward stages."""
import pytest

from _grad_cases import GRAD_CASES

TOL = 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", GRAD_CASES)
def test_training_stages_teacher_forced(name):
    from _stage_checks import run
    lines = []
    out, errs, flat, gap = run(name, log=lines.append)
    print("\n" + "\n".join(lines))
    assert any(k.startswith("bwd layer") for k in out) and any(k.startswith("fwd layer") for k in out)
    # lin3's parameter gradients are sums over ~2 A H terms that largely cancel (the whole-step test gates them against
    # the reference's own float32 gap): float32 torch autograd and the float32 kernel agree to a few 1e-5 there
    bad = {k: v for k, v in out.items() if max(v) > (5e-5 if k.startswith("bwd scalarize") else TOL)}
    assert not bad, bad


@pytest.mark.gpu
def test_stage_entries_order_their_own_streams_against_the_caller():
    """The ordering bug of round 4 (a stage entry returned with its weight-gradient stream still running; the caller's next
    writes raced with it - 4 failures in 16 runs, found by luck), made deterministic: the whole teacher-forced sequence runs on a
    NON-default stream, and right after every oard_train_stage_backward call the cotangent tensors it was given are overwritten
    with NaN on that stream.  Every gradient must still pass its gate - a stage that returns before its second stream has
    consumed its inputs turns them into NaN."""
    import torch
    from _stage_checks import run
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    lines = []
    with torch.cuda.stream(side):
        out, errs, flat, gap = run("g9_grad_h32", log=lines.append, clobber=True)
    side.synchronize()
    assert any(k.startswith("bwd layer") for k in out)
    bad = {k: v for k, v in out.items() if not all(x == x for x in v) or max(v) > (5e-5 if k.startswith("bwd scalarize") else TOL)}
    assert not bad, bad
