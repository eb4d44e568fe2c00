"""Learning-rate schedules with the reference's names and argument meaning (utils/lrschedule.py:5-35), as plain
host scalars: the value is written into the device-side Adam state with `FusedAdam.set_lr`, so a captured hipGraph
step picks it up on its next replay without re-capture."""
import math


def noam_learning_rate_decay(init_lr, global_step, warmup_steps=2000):
    """utils/lrschedule.py:5-11 (Noam scheme of tensor2tensor)."""
    warmup_steps = float(warmup_steps)
    step = global_step + 1.0
    return init_lr * warmup_steps ** 0.5 * min(step * warmup_steps ** -1.5, step ** -0.5)


def step_learning_rate_decay(init_lr, global_step, anneal_rate=0.98, anneal_interval=50000):
    """utils/lrschedule.py:14-17."""
    return init_lr * anneal_rate ** (global_step // anneal_interval)


def cyclic_cosine_annealing(init_lr, global_step, T, M):
    """utils/lrschedule.py:20-35 (T total iterations, M cycles)."""
    TdivM = T // M
    return init_lr / 2.0 * (math.cos(math.pi * ((global_step - 1) % TdivM) / TdivM) + 1.0)


def apply_schedule(optimizer, schedule, init_lr, global_step, **kw):
    """optimizer: model.FusedAdam (or anything with set_lr); returns the lr that was set."""
    lr = float(schedule(init_lr, global_step, **kw))
    optimizer.set_lr(lr)
    return lr
