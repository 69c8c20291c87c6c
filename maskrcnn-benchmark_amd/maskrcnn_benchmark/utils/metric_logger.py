"""SmoothedValue / MetricLogger (reference utils/metric_logger.py:8-66).

Values may be device tensors: they are stored as-is and only converted to Python floats when a
statistic is read (i.e. when a log line is printed), so that logging does not force a host sync
every iteration the way the reference's `.item()` calls do (SURVEY.md App. C "logging")."""
from collections import defaultdict, deque

import torch


class SmoothedValue(object):
    def __init__(self, window_size=20):
        self.deque = deque(maxlen=window_size)
        self._total = 0.0
        self._count = 0
        self._pending = []

    def update(self, value):
        self._pending.append(value)
        if len(self._pending) > 4096:
            self._drain()

    def _drain(self):
        for v in self._pending:
            v = float(v.item()) if isinstance(v, torch.Tensor) else float(v)
            self.deque.append(v)
            self._count += 1
            self._total += v
        self._pending = []

    @property
    def count(self):      # the reference's public counters (utils/metric_logger.py:16-17), exact after every update
        self._drain()
        return self._count

    @property
    def total(self):
        self._drain()
        return self._total

    @property
    def median(self):
        self._drain()
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        self._drain()
        return torch.tensor(list(self.deque)).mean().item()

    @property
    def global_avg(self):
        self._drain()
        return self._total / max(self._count, 1)


class MetricLogger(object):
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            assert isinstance(v, (float, int, torch.Tensor))
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError("'{}' object has no attribute '{}'".format(type(self).__name__, attr))

    def __str__(self):
        return self.delimiter.join("{}: {:.4f} ({:.4f})".format(n, m.median, m.global_avg)
                                   for n, m in self.meters.items())
