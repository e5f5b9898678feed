"""Iterate-callback criteria and the best-iterate store behind `use_opti_callback` (what hippopt.base.opti_callback provides:
base/opti_callback.py:24-249 the criteria, :252-373 SaveBestUnsolvedVariablesCallback; used by the fallback of OptiSolver.solve,
base/opti_solver.py:451-520).

Specification (the reference's behaviour, restated):
  * a criterion looks at the current iterate and answers `satisfied()`; `update()` is called only when the WHOLE criterion a store
    was built with is satisfied, and moves the criterion's running record; `reset()` forgets the record;
  * four scalar criteria over two quantities of an iterate — cost and primal infeasibility:
        BestCost / BestPrimalInfeasibility         satisfied when the quantity is below the running best; update: best := quantity
        AcceptableCost / AcceptablePrimalInfeasibility(threshold)
                                                   satisfied when the quantity is below the fixed threshold; update: the best
                                                   acceptable value seen so far (a record only, it does not enter `satisfied`)
  * `a & b`, `a | b` combine criteria (anything else is a TypeError); a combination forwards iterate, update and reset to both sides;
  * the store keeps x (and, when asked, the per-term costs and the constraint multipliers) of the last iterate that satisfied its
    criterion.
The reference's criteria pull the cost from `opti.debug.value(opti.f)` and the infeasibility from IPOPT's
`stats()["iterations"]["inf_pr"][-1]`; here the NLP driver hands both over as an `IterateInfo` at every iteration (IPOPT's
intermediate callback / SciPy's `callback(xk, state)`).  One parametrised class stands behind the four scalar criteria and one behind
the two combinations; the public class and attribute names are the reference's.
"""
import abc
import logging
import operator
from typing import NamedTuple

import numpy as np


class IterateInfo(NamedTuple):
    iteration: int
    cost: float
    primal_infeasibility: float


class CallbackCriterion(abc.ABC):
    info = None      # the iterate under consideration (update_iterate)

    @abc.abstractmethod
    def satisfied(self) -> bool: ...

    @abc.abstractmethod
    def update(self) -> None: ...

    @abc.abstractmethod
    def reset(self) -> None: ...

    def update_iterate(self, info: IterateInfo) -> None:
        """what update_opti_debug does for the reference's criteria (opti_callback.py:76-78)"""
        self.info = info

    def _combined(self, other, how):
        if not isinstance(other, CallbackCriterion):
            raise TypeError("a criterion combines with another criterion, not with " + type(other).__name__)
        return how(lhs=self, rhs=other)

    def __and__(self, other):
        return self._combined(other, AndCombinedCallbackCriterion)

    def __or__(self, other):
        return self._combined(other, OrCombinedCallbackCriterion)

    __rand__, __ror__ = __and__, __or__


class _ScalarCriterion(CallbackCriterion):
    """quantity < bound, where the bound is either the running record itself (Best*) or a fixed threshold (Acceptable*).
    Subclasses name the quantity, the record attribute and — for the fixed kind — the threshold attribute."""
    quantity = record = threshold = None

    def __init__(self, *fixed, **named) -> None:
        if self.threshold is not None:
            value = fixed[0] if fixed else named.get(self.threshold, np.inf)
            setattr(self, self.threshold, value)
        self.reset()

    def _now(self):
        return getattr(self.info, self.quantity)

    def reset(self) -> None:
        setattr(self, self.record, np.inf)

    def satisfied(self) -> bool:
        return self._now() < getattr(self, self.threshold or self.record)

    def update(self) -> None:
        old, new = getattr(self, self.record), self._now()
        if self.threshold is None or new < old:
            logging.getLogger("[hippopt_amd::%s]" % type(self).__name__).debug("%s: %s -> %s", self.record, old, new)
            setattr(self, self.record, new)


class BestCost(_ScalarCriterion):
    quantity, record = "cost", "best_cost"


class BestPrimalInfeasibility(_ScalarCriterion):
    quantity, record = "primal_infeasibility", "best_primal_infeasibility"


class AcceptableCost(_ScalarCriterion):
    quantity, record, threshold = "cost", "best_acceptable_cost", "acceptable_cost"


class AcceptablePrimalInfeasibility(_ScalarCriterion):
    quantity, record, threshold = "primal_infeasibility", "best_acceptable_primal_infeasibility", "acceptable_primal_infeasibility"


class CombinedCallbackCriterion(CallbackCriterion):
    connective = None    # operator.and_ / operator.or_ on the two verdicts

    def __init__(self, lhs: CallbackCriterion, rhs: CallbackCriterion) -> None:
        self.lhs, self.rhs = lhs, rhs

    def _both(self, method, *args):
        for side in (self.lhs, self.rhs):
            getattr(side, method)(*args)

    def satisfied(self) -> bool:
        return bool(self.connective(self.lhs.satisfied(), self.rhs.satisfied()))

    def update(self) -> None:
        self._both("update")

    def reset(self) -> None:
        self._both("reset")

    def update_iterate(self, info: IterateInfo) -> None:
        self._both("update_iterate", info)


class AndCombinedCallbackCriterion(CombinedCallbackCriterion):
    connective = staticmethod(operator.and_)


class OrCombinedCallbackCriterion(CombinedCallbackCriterion):
    connective = staticmethod(operator.or_)


class SaveBestUnsolvedVariablesCallback:
    """x, cost and — when asked — per-term costs and constraint multipliers of the iterate that last satisfied the criterion
    (opti_callback.py:310-373)."""

    def __init__(self, criterion: CallbackCriterion, save_costs: bool = True, save_constraint_multipliers: bool = True) -> None:
        criterion.reset()
        self.criterion = criterion
        self.save_costs, self.save_constraint_multipliers = save_costs, save_constraint_multipliers
        self.best_iteration = self.best_x = self.best_cost = self.best_constraint_multipliers = None
        self.best_cost_values = {}

    def call(self, info: IterateInfo, x, multipliers=None, cost_values=None) -> None:
        crit = self.criterion
        crit.update_iterate(info)
        if not crit.satisfied():
            return
        crit.update()
        logging.getLogger("[hippopt_amd::SaveBestUnsolvedVariablesCallback]").info("[i=%d] New best intermediate variables", info.iteration)
        self.best_iteration, self.best_cost = info.iteration, info.cost
        self.best_x = np.array(x, dtype=float)
        if cost_values is not None and self.save_costs:
            self.best_cost_values = dict(cost_values() if callable(cost_values) else cost_values)
        if multipliers is not None and self.save_constraint_multipliers:
            self.best_constraint_multipliers = np.array(multipliers, dtype=float)

    __call__ = call
