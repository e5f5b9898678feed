"""Iterate-callback criteria and the best-iterate store: mirror of hippopt.base.opti_callback (reference:
base/opti_callback.py:24-249 criteria, :252-373 SaveBestUnsolvedVariablesCallback) and of the fallback in OptiSolver.solve
(base/opti_solver.py:451-520).

The reference's criteria read the current cost from `opti.debug.value(opti.f)` and the primal infeasibility from IPOPT's
`stats()["iterations"]["inf_pr"][-1]`; here the NLP driver hands both over as an `IterateInfo` at every iteration (IPOPT's
intermediate callback / SciPy's `callback(xk, state)`), the class names, combination operators and update rules are the same.
"""
import abc
import dataclasses
import logging

import numpy as np


@dataclasses.dataclass
class IterateInfo:
    iteration: int
    cost: float
    primal_infeasibility: float


class CallbackCriterion(abc.ABC):
    def __init__(self) -> None:
        self.info = None

    @abc.abstractmethod
    def satisfied(self) -> bool:
        pass

    @abc.abstractmethod
    def update(self) -> None:
        pass

    @abc.abstractmethod
    def reset(self) -> None:
        pass

    def __or__(self, other):
        if not isinstance(other, CallbackCriterion):
            raise TypeError(other)
        return OrCombinedCallbackCriterion(lhs=self, rhs=other)

    def __ror__(self, other):
        return self.__or__(other)

    def __and__(self, other):
        if not isinstance(other, CallbackCriterion):
            raise TypeError(other)
        return AndCombinedCallbackCriterion(lhs=self, rhs=other)

    def __rand__(self, other):
        return self.__and__(other)

    def update_iterate(self, info: IterateInfo) -> None:
        """Counterpart of update_opti_debug (opti_callback.py:76-78)."""
        self.info = info


class BestCost(CallbackCriterion):
    def __init__(self) -> None:
        CallbackCriterion.__init__(self)
        self.best_cost = None
        self.reset()

    def reset(self) -> None:
        self.best_cost = np.inf

    def satisfied(self) -> bool:
        return self.info.cost < self.best_cost

    def update(self) -> None:
        logging.getLogger("[hippopt_amd::BestCost]").debug(f"New best cost: {self.info.cost} (old: {self.best_cost})")
        self.best_cost = self.info.cost


class AcceptableCost(CallbackCriterion):
    def __init__(self, acceptable_cost: float = np.inf) -> None:
        CallbackCriterion.__init__(self)
        self.acceptable_cost = acceptable_cost
        self.best_acceptable_cost = None
        self.reset()

    def reset(self) -> None:
        self.best_acceptable_cost = np.inf

    def satisfied(self) -> bool:
        return self.info.cost < self.acceptable_cost

    def update(self) -> None:
        if self.info.cost < self.best_acceptable_cost:
            self.best_acceptable_cost = self.info.cost


class AcceptablePrimalInfeasibility(CallbackCriterion):
    def __init__(self, acceptable_primal_infeasibility: float = np.inf) -> None:
        CallbackCriterion.__init__(self)
        self.acceptable_primal_infeasibility = acceptable_primal_infeasibility
        self.best_acceptable_primal_infeasibility = None
        self.reset()

    def reset(self) -> None:
        self.best_acceptable_primal_infeasibility = np.inf

    def satisfied(self) -> bool:
        return self.info.primal_infeasibility < self.acceptable_primal_infeasibility

    def update(self) -> None:
        if self.info.primal_infeasibility < self.best_acceptable_primal_infeasibility:
            self.best_acceptable_primal_infeasibility = self.info.primal_infeasibility


class BestPrimalInfeasibility(CallbackCriterion):
    def __init__(self) -> None:
        CallbackCriterion.__init__(self)
        self.best_primal_infeasibility = None
        self.reset()

    def reset(self) -> None:
        self.best_primal_infeasibility = np.inf

    def satisfied(self) -> bool:
        return self.info.primal_infeasibility < self.best_primal_infeasibility

    def update(self) -> None:
        self.best_primal_infeasibility = self.info.primal_infeasibility


class CombinedCallbackCriterion(CallbackCriterion, abc.ABC):
    def __init__(self, lhs: CallbackCriterion, rhs: CallbackCriterion) -> None:
        CallbackCriterion.__init__(self)
        self.lhs, self.rhs = lhs, rhs

    def reset(self) -> None:
        self.lhs.reset()
        self.rhs.reset()

    def update(self) -> None:
        self.lhs.update()
        self.rhs.update()

    def update_iterate(self, info: IterateInfo) -> None:
        self.lhs.update_iterate(info)
        self.rhs.update_iterate(info)


class OrCombinedCallbackCriterion(CombinedCallbackCriterion):
    def satisfied(self) -> bool:
        return self.lhs.satisfied() or self.rhs.satisfied()


class AndCombinedCallbackCriterion(CombinedCallbackCriterion):
    def satisfied(self) -> bool:
        return self.lhs.satisfied() and self.rhs.satisfied()


class SaveBestUnsolvedVariablesCallback:
    """Keeps the iterate that last satisfied the criterion (opti_callback.py:310-373): x, cost, and — when asked — the
    per-term costs and the constraint multipliers at that iterate."""

    def __init__(self, criterion: CallbackCriterion, save_costs: bool = True, save_constraint_multipliers: bool = True) -> None:
        self.criterion = criterion
        self.criterion.reset()
        self.save_costs, self.save_constraint_multipliers = save_costs, save_constraint_multipliers
        self.best_iteration = None
        self.best_x = None
        self.best_cost = None
        self.best_cost_values = {}
        self.best_constraint_multipliers = None

    def __call__(self, info: IterateInfo, x, multipliers=None, cost_values=None) -> None:
        self.call(info, x, multipliers, cost_values)

    def call(self, info: IterateInfo, x, multipliers=None, cost_values=None) -> None:
        self.criterion.update_iterate(info)
        if self.criterion.satisfied():
            self.criterion.update()
            logging.getLogger("[hippopt_amd::SaveBestUnsolvedVariablesCallback]").info(f"[i={info.iteration}] New best intermediate variables")
            self.best_iteration = info.iteration
            self.best_cost = info.cost
            self.best_x = np.array(x, dtype=float, copy=True)
            if self.save_costs and cost_values is not None:
                self.best_cost_values = dict(cost_values() if callable(cost_values) else cost_values)
            if self.save_constraint_multipliers and multipliers is not None:
                self.best_constraint_multipliers = np.array(multipliers, dtype=float, copy=True)
