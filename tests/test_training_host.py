"""Host-side pieces of the training step that need no GPU: the learning-rate schedule against the installed
transformers implementation the reference calls (train.py:242-246)."""
import pytest


def test_cosine_schedule_matches_transformers():
    tr = pytest.importorskip("transformers")
    import torch
    from kosmosx.training import cosine_schedule_with_warmup
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    total, warm = 200, int(200 * 0.01)                 # NUM_WARMUP_STEPS = int(max_train_steps * 0.01)
    sched = tr.get_cosine_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total)
    for step in range(total + 5):
        assert abs(sched.get_last_lr()[0] - cosine_schedule_with_warmup(step, warm, total)) < 1e-12, step
        opt.step(); sched.step()
