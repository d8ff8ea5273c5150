"""Oracle: OneCycleLR (linear anneal, no momentum cycling) as configure_optimizers builds it,
modules/detection.py:485-518.  TEST INFRASTRUCTURE.

The reference passes ``final_div_factor / div_factor`` to torch so that the final lr is
``max_lr / final_div_factor`` (config/general.yaml:16-17).  Closed form of
torch.optim.lr_scheduler.OneCycleLR with anneal_strategy='linear', three_phase=False."""


def one_cycle_lr(step, max_lr, total_steps, pct_start, div_factor, final_div_factor):
    initial_lr = max_lr / div_factor
    min_lr = initial_lr / (final_div_factor / div_factor)
    end1 = float(pct_start * total_steps) - 1
    end2 = total_steps - 1
    if step <= end1:
        pct = step / end1
        return (max_lr - initial_lr) * pct + initial_lr
    pct = (step - end1) / (end2 - end1)
    return (min_lr - max_lr) * pct + max_lr
