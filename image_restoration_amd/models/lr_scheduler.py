"""Learning-rate schedules of the path, restated without torch's _LRScheduler machinery.

Same sequences as the reference ``basicsr/models/lr_scheduler.py`` (MultiStepRestartLR :6-33,
CosineAnnealingRestartLR :57-96) when driven the way BaseModel.update_learning_rate drives them
(step() once per iteration from iteration 2 on, base_model.py:154-156); pinned by golden G-j.
"""
import math
from collections import Counter


class _Scheduler:
    """Minimal counterpart of torch.optim.lr_scheduler._LRScheduler(last_epoch=-1): construction records
    ``initial_lr`` in every param group and performs the initial step (last_epoch 0)."""

    def __init__(self, optimizer):
        self.optimizer = optimizer
        for g in optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_lrs = [g['initial_lr'] for g in optimizer.param_groups]
        self.last_epoch = -1
        self.step()

    def step(self):
        self.last_epoch += 1
        for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
            g['lr'] = lr

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


class MultiStepRestartLR(_Scheduler):
    """MultiStepRestartLR(optimizer, milestones, gamma=0.1, restarts=(0,), restart_weights=(1,))."""

    def __init__(self, optimizer, milestones, gamma=0.1, restarts=(0, ), restart_weights=(1, ), last_epoch=-1):
        self.milestones = Counter(milestones)
        self.gamma = gamma
        self.restarts = list(restarts)
        self.restart_weights = list(restart_weights)
        assert len(self.restarts) == len(self.restart_weights), 'restarts and their weights do not match.'
        super().__init__(optimizer)

    def get_lr(self):
        groups = self.optimizer.param_groups
        if self.last_epoch in self.restarts:
            weight = self.restart_weights[self.restarts.index(self.last_epoch)]
            return [g['initial_lr'] * weight for g in groups]
        if self.last_epoch not in self.milestones:
            return [g['lr'] for g in groups]
        return [g['lr'] * self.gamma**self.milestones[self.last_epoch] for g in groups]


def get_position_from_periods(iteration, cumulative_period):
    """Index of the first cumulative period >= iteration (reference :36-54)."""
    for i, period in enumerate(cumulative_period):
        if iteration <= period:
            return i


class CosineAnnealingRestartLR(_Scheduler):
    """CosineAnnealingRestartLR(optimizer, periods, restart_weights=(1,), eta_min=0)."""

    def __init__(self, optimizer, periods, restart_weights=(1, ), eta_min=0, last_epoch=-1):
        self.periods = list(periods)
        self.restart_weights = list(restart_weights)
        self.eta_min = eta_min
        assert len(self.periods) == len(self.restart_weights), 'periods and restart_weights should have the same length.'
        self.cumulative_period = [sum(self.periods[0:i + 1]) for i in range(len(self.periods))]
        super().__init__(optimizer)

    def get_lr(self):
        idx = get_position_from_periods(self.last_epoch, self.cumulative_period)
        w = self.restart_weights[idx]
        nearest_restart = 0 if idx == 0 else self.cumulative_period[idx - 1]
        period = self.periods[idx]
        return [self.eta_min + w * 0.5 * (base - self.eta_min) *
                (1 + math.cos(math.pi * ((self.last_epoch - nearest_restart) / period))) for base in self.base_lrs]
