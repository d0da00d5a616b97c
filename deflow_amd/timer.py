"""Nested wall-clock timer with the call surface of ``dztimer.Timing`` as DeFlow uses it
([REF deflow.py:38-39,55-95]): ``Timing()``, ``.start(name)``, ``.stop()``, ``timer[i]`` -> child, nestable.
Like the original it measures host time (kernel launches, not kernel execution) unless ``sync=True``."""
from __future__ import annotations

import time
from typing import Dict


class Timing:
    def __init__(self, sync: bool = False):
        self.name = ""
        self.total = 0.0
        self.count = 0
        self._t0 = None
        self._children: Dict[int, "Timing"] = {}
        self._sync = sync

    def start(self, name: str = ""):
        if name:
            self.name = name
        if self._sync:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        self._t0 = time.perf_counter()

    def stop(self):
        if self._t0 is None:
            return
        if self._sync:
            import torch
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        self.total += time.perf_counter() - self._t0
        self.count += 1
        self._t0 = None

    def __getitem__(self, i: int) -> "Timing":
        if i not in self._children:
            self._children[i] = Timing(self._sync)
        return self._children[i]

    def report(self, indent: int = 0) -> str:
        lines = [f"{'  ' * indent}{self.name or '<unnamed>'}: {self.total * 1e3:.3f} ms over {self.count} calls"]
        for k in sorted(self._children):
            lines.append(self._children[k].report(indent + 1))
        return "\n".join(lines)

    def print(self, *_a, **_k):
        print(self.report())
