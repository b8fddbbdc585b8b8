"""Learning-rate and sequence-length schedules of the training driver — behavioural mirror of
lcasr/utils/scheduling.py (CosineLRScheduler :3-28, SequenceWarmupManager :32-100): same constructor arguments, same
state_dict contents, same step() results (pinned by tests/golden/schedules.npz, generated from the reference classes)."""
from __future__ import annotations

import math

import torch


class CosineLRScheduler(torch.optim.lr_scheduler._LRScheduler):
    """Linear warm-up to `peak_value` over `warmup_steps` scheduler steps, then (after `set_cosine_schedule`) a half cosine
    from peak to `final_value` over the remaining recordings."""

    def __init__(self, optimizer, warmup_steps, peak_value, final_value):
        self.is_warmup = True
        self.warmup_steps = warmup_steps
        self.peak_value = peak_value
        self.final_value = final_value
        self.offset = 0
        super().__init__(optimizer)

    def is_warming_up(self):
        return self.is_warmup and self.last_epoch < self.warmup_steps

    def set_cosine_schedule(self, total_recordings, cur_podcast):
        self.last_epoch = 0                                      # the cosine phase counts from here
        self.is_warmup = False
        self.steps = total_recordings - cur_podcast + 1
        self.offset = -cur_podcast

    def get_lr(self):
        if self.is_warmup:
            lr = self.peak_value * min(1.0, self.last_epoch / self.warmup_steps)
        else:
            phase = (self.last_epoch + self.offset) / self.steps * math.pi
            lr = self.final_value + 0.5 * (self.peak_value - self.final_value) * (1 + math.cos(phase))
        return [lr for _ in self.base_lrs]


class SequenceWarmupManager:
    """Sequence-length curriculum: every `increase_every` recordings (after `start_after`) the chunk length is multiplied by
    `increase_by_multiplier` (capped at `max_sequence_length`) and the batch size by `batch_size_multiplier`."""

    def __init__(self, increase_every: int, stop_after: int, start_after: int, initial_sequence_length: int,
                 initial_batch_size: int, max_sequence_length: int, increase_by_multiplier: float = 2.0,
                 batch_size_multiplier: float = 0.5, cur_position: int = 0, steps_since_last_increase: int = 0, **kwargs):
        self.increase_every = increase_every                     # -1 disables the curriculum
        self.stop_after = stop_after
        self.start_after = start_after
        self.max_sequence_length = max_sequence_length
        self.increase_by_multiplier = increase_by_multiplier
        self.cur_position = cur_position
        self.batch_size_multiplier = batch_size_multiplier
        self.cur_sequence_length = initial_sequence_length
        self.cur_batch_size = initial_batch_size
        self.steps_since_last_increase = steps_since_last_increase

    def _same(self):
        return False, self.cur_sequence_length, self.cur_batch_size

    def _grow(self, next_len):
        self.steps_since_last_increase = 0
        self.cur_sequence_length = next_len
        self.cur_batch_size = max(int(self.cur_batch_size * self.batch_size_multiplier), 1)
        return True, self.cur_sequence_length, self.cur_batch_size

    def step(self, steps=1):
        """Advance by `steps` recordings; returns (changed, sequence_length, batch_size)."""
        if self.increase_every == -1:
            return self._same()
        next_len = max(int(self.cur_sequence_length * self.increase_by_multiplier), 1)
        self.cur_position += steps
        past_stop = self.cur_position >= self.stop_after
        half_way = self.steps_since_last_increase >= self.increase_every / 2
        if self.cur_position < self.start_after:
            return self._same()
        if past_stop and not half_way:
            return self._same()
        if self.cur_sequence_length * self.increase_by_multiplier > self.max_sequence_length:
            if self.cur_sequence_length == self.max_sequence_length:
                return self._same()
            next_len = self.max_sequence_length                  # final, partial increase: falls through to the counter below
        elif past_stop and half_way:
            return self._grow(next_len)                          # one last doubling if at least half an interval has passed
        self.steps_since_last_increase += steps
        if self.steps_since_last_increase >= self.increase_every:
            return self._grow(next_len)
        return self._same()

    def state_dict(self):
        return self.__dict__

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)
