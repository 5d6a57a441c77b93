"""Learning-rate schedules of the SFT loop (host logic, no kernels).

Mirrors the names and parameters of ``finetrainers/optimizer.py:191-226`` (``get_lr_scheduler`` -> ``torch.optim.lr_scheduler.LambdaLR`` over
one of seven multiplier functions) for the fused optimiser step, which takes its learning rate as a kernel argument instead of reading a
``torch.optim`` param group.  ``LRSchedule`` follows LambdaLR's clock: the rate in force before any ``step()`` is ``base_lr * f(0)`` (0 with a
warm-up), and the k-th call of ``step()`` moves it to ``base_lr * f(k)`` -- the reference calls it once per optimiser step, after
``optimizer.step()`` (trainer/sft_trainer/trainer.py:500-503).  Pinned against the reference's own functions in tests/golden (``lr.*``).
"""

from __future__ import annotations

import math
from typing import Any, Callable, Dict, Optional


def _warmup(step: int, warmup: int) -> Optional[float]:
    return step / max(1, warmup) if step < warmup else None


def lr_multiplier(name: str, num_warmup_steps: int = 0, num_training_steps: int = 0, num_cycles: float = 1, power: float = 1.0,
                  lr_init: float = 1e-3, lr_end: float = 1e-7, step_rules: Optional[str] = None) -> Callable[[int], float]:
    """step -> multiplier of the base learning rate, for the schedule names ``--lr_scheduler`` accepts."""
    name = name.lower()
    w, n = int(num_warmup_steps or 0), int(num_training_steps or 0)
    if name == "constant":
        return lambda step: 1.0
    if name == "constant_with_warmup":
        return lambda step: (lambda u: 1.0 if u is None else u)(_warmup(step, w))
    if name == "piecewise_constant":
        # "m0:s0,m1:s1,...,m_last": multiplier m_i while step < s_i (boundaries in ascending order), m_last afterwards
        parts = step_rules.split(",")
        table = sorted((int(p.split(":")[1]), float(p.split(":")[0])) for p in parts[:-1])
        tail = float(parts[-1])

        def f(step: int) -> float:
            for bound, mult in table:
                if step < bound:
                    return mult
            return tail
        return f
    if name == "linear":
        def f(step: int) -> float:
            u = _warmup(step, w)
            return u if u is not None else max(0.0, (n - step) / max(1, n - w))
        return f
    if name == "cosine":
        def f(step: int) -> float:
            u = _warmup(step, w)
            if u is not None:
                return u
            progress = (step - w) / max(1, n - w)
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
        return f
    if name == "cosine_with_restarts":
        def f(step: int) -> float:
            u = _warmup(step, w)
            if u is not None:
                return u
            progress = (step - w) / max(1, n - w)
            if progress >= 1.0:
                return 0.0
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(num_cycles) * progress) % 1.0))))
        return f
    if name == "polynomial":
        if not lr_init > lr_end:
            raise ValueError(f"lr_end ({lr_end}) must be smaller than the initial lr ({lr_init})")

        def f(step: int) -> float:
            u = _warmup(step, w)
            if u is not None:
                return u
            if step > n:
                return lr_end / lr_init
            remaining = 1 - (step - w) / (n - w)
            return ((lr_init - lr_end) * remaining ** power + lr_end) / lr_init
        return f
    raise ValueError(f"Unsupported scheduler: {name}")


class LRSchedule:
    """``LambdaLR``-clocked learning rate for ``MI355XSFTStep`` (``get_last_lr`` / ``step`` / ``state_dict`` like the torch scheduler)."""

    def __init__(self, base_lr: float, multiplier: Callable[[int], float], last_epoch: int = -1):
        self.base_lr, self.multiplier = float(base_lr), multiplier
        self.last_epoch = last_epoch + 1  # LambdaLR performs one initial step at construction

    @classmethod
    def from_args(cls, base_lr: float, name: str, **kwargs) -> "LRSchedule":
        # lr_init / lr_end of the polynomial schedule keep the reference's defaults (1e-3, 1e-7: optimizer.py:200-201) unless the caller passes
        # them -- the reference trainer never does (sft_trainer/trainer.py:221-228), so its multiplier decays to 1e-4 whatever the base lr
        return cls(base_lr, lr_multiplier(name, **kwargs))

    def current_lr(self) -> float:
        return self.base_lr * self.multiplier(self.last_epoch)

    def get_last_lr(self):
        return [self.current_lr()]

    def step(self) -> None:
        self.last_epoch += 1

    def state_dict(self) -> Dict[str, Any]:
        return {"last_epoch": self.last_epoch, "base_lrs": [self.base_lr]}

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        self.last_epoch = int(sd["last_epoch"])
        if "base_lrs" in sd:
            self.base_lr = float(sd["base_lrs"][0])
