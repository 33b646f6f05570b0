"""Learning-rate schedules (host-side scalars; reference optim/sched.py)."""
from math import ceil


def noam_schedule(step, warmup_step=4000):
    if step <= warmup_step:
        return step / warmup_step
    return (warmup_step ** 0.5) * (step ** -0.5)


def warmup_linear(step, warmup_step, tot_step):
    if step < warmup_step:
        return step / warmup_step
    return max(0, (tot_step - step) / (tot_step - warmup_step))


def vqa_schedule(step, warmup_interval, decay_interval, decay_start, decay_rate):
    if step < warmup_interval:
        return 1 / 4
    elif step < 2 * warmup_interval:
        return 2 / 4
    elif step < 3 * warmup_interval:
        return 3 / 4
    elif step >= decay_start:
        return decay_rate ** ceil((step - decay_start) / decay_interval)
    return 1


def _sched(base_lr, global_step, opts):
    if opts.decay == 'linear':
        lr = base_lr * warmup_linear(global_step, opts.warmup_steps, opts.num_train_steps)
    elif opts.decay == 'invsqrt':
        lr = base_lr * noam_schedule(global_step, opts.warmup_steps)
    elif opts.decay == 'constant':
        lr = base_lr
    elif opts.decay == 'vqa':
        lr = base_lr * vqa_schedule(global_step, opts.warm_int, opts.decay_int, opts.decay_st, opts.decay_rate)
    else:
        raise ValueError('invalid decay %r' % (opts.decay,))
    return lr if lr > 0 else 1e-8          # safeguard against a miscounted number of train steps


def get_lr_sched(global_step, opts):
    return _sched(opts.learning_rate, global_step, opts)


def get_xlmr_lr_sched(global_step, opts):
    return _sched(opts.xlmr_lr, global_step, opts)
