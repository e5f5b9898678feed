"""`ExpressionType` and `Output` of hippopt's problem facade (reference: base/problem.py:15-79)."""
import dataclasses
from enum import Enum
from typing import Any


class ExpressionType(Enum):
    skip = 0
    subject_to = 1
    minimize = 2


class ProblemNotSolvedException(Exception):
    def __init__(self):
        super().__init__("No solution is available. Was solve() called successfully?")


@dataclasses.dataclass
class Output:
    values: Any = None
    cost_value: float = None
    cost_values: dict = dataclasses.field(default_factory=dict)
    constraint_multipliers: dict = dataclasses.field(default_factory=dict)

    @staticmethod
    def _nest(flat: dict) -> dict:
        """{"a.b": v} -> {"a": {"b": v}}  (problem.py:58-79: keys are split at '.')."""
        out: dict = {}
        for key, value in flat.items():
            node = out
            parts = key.split(".")
            for part in parts[:-1]:
                node = node.setdefault(part, {})
            node[parts[-1]] = value
        return out

    def to_dict(self) -> dict:
        values = self.values
        if isinstance(values, list):
            values = [v.to_dict(flatten=False) for v in values]
        elif values is not None:
            values = values.to_dict(flatten=False)
        return {"values": values, "cost_value": self.cost_value, "cost_values": Output._nest(self.cost_values),
                "constraint_multipliers": Output._nest(self.constraint_multipliers)}
