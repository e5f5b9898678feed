"""HipNlpSolver — the solver plugin that stands where hippopt's `OptiSolver` stands (reference: base/opti_solver.py:49-638,
abstract interface base/optimization_solver.py:25-96), for the kinodynamic NLP.

`OptiSolver` receives symbolic `cs.MX` costs/constraints and lets CasADi evaluate them on one CPU thread.  This plugin owns
the *typed* kinodynamic problem instead: the flat layout of the structure (same names / order as the reference), the
parameter vector, and an engine handle (`HipNlp`, libhipnlp.so) that evaluates f, grad f, g, jac g of all knots in one HIP
launch per callback.  `add_cost` / `add_constraint` with symbolic expressions are therefore not available (the reference's
planner list is built into the engine); everything else keeps the reference's method names and semantics.

The NLP driver: IPOPT (through `cyipopt`) when importable; otherwise SciPy's `trust-constr` as a stand-in for small horizons
(IPOPT is not installed in the build image).  Both consume exactly the IPOPT callback quartet of INTEGRATION.md.
"""
import copy
import logging

import numpy as np

from . import _abi
from .base import OptimizationObject, extend_structure_to_horizon
from .base.optimal_control import OptimizationSolver, TypedProblemError
from .base.opti_callback import CallbackCriterion, IterateInfo, SaveBestUnsolvedVariablesCallback
from .base.optimization_object import STORAGE_TYPE
from .hipnlp import HipNlpError


class HipFailure(Exception):
    """Counterpart of OptiFailure (opti_solver.py:28-37)."""

    def __init__(self, message, callback_used=False):
        info = " and the callback did not manage to save an intermediate solution" if callback_used else ""
        super().__init__(f"The NLP solver failed to solve the problem{info}. Message: {message}")


class InitialGuessFailure(Exception):
    def __init__(self, message):
        super().__init__("Failed to set the initial guess. Message: " + str(message))


class _PoseEngine:
    """HipPose behind the method set the NLP drivers use on HipNlp (single pose)."""

    def __init__(self, pose):
        self._pose = pose
        self.n, self.m, self.nnz = pose.n, pose.m, pose.nnz

    def set_params(self, p):
        self._pose.set_params(p)

    def bounds(self):
        lb, ub = self._pose.bounds()
        return np.full(self.n, -np.inf), np.full(self.n, np.inf), lb[0], ub[0]

    def sparsity(self):
        return self._pose.sparsity()

    def eval(self, x, new_x=True, want=("f", "grad", "g", "jac"), nan_ok=False):
        try:
            return self._pose.eval(x)   # (81 variables: always evaluated)
        except HipNlpError as err:
            if not (nan_ok and err.code == -5):
                raise
            nan = np.full
            return nan(1, np.nan), nan((1, self.n), np.nan), nan((1, self.m), np.nan), nan((1, self.nnz), np.nan)

    def cost_terms(self):
        return self._pose.cost_terms()

    def row_blocks(self):
        return [(name, first, rows, 0, 1) for name, first, rows in self._pose.row_blocks()]

    # exact Hessian of the Lagrangian (the reference pose finder runs IPOPT with its default exact-Hessian option,
    # humanoid_pose_finder/main.py:101): lower triangle (row >= col)
    def hess_sparsity(self):
        return self._pose.hess_sparsity()

    def eval_hess(self, x, obj_factor, lam):
        return self._pose.eval_hess(x, obj_factor, lam)[0]


class _SimpleBoundsLift:
    """The NLP as nlpsol's `detect_simple_bounds` presents it to IPOPT (the reference runs Opti with {"expand": True,
    "detect_simple_bounds": True}, main_periodic_step.py:109-110): every row of g that is exactly ONE decision variable — the u_v and
    joint position / velocity boxes of planner.py:386-405,699-719 (70 rows per interior knot), the `x_0 == initial_state` rows and the
    final-state rows — is a bound on that variable (lbx / ubx), not a row of g.
    The engine handle itself IS that reduced problem (HIPNLP_FLAG_DETECT_SIMPLE_BOUNDS: the kernels neither compute nor move the
    lifted rows; sizes, pattern, bounds, g, jac g and the Hessian's multipliers are the reduced ones): nothing is gathered on the
    host.  What is left here is the way BACK for the multipliers, so that `Output.constraint_multipliers` still carries every named
    constraint: the multiplier of a lifted row is the bound multiplier of its variable (z_U - z_L), given to the row whose bound is
    the active one (`hipnlp_lift_map`)."""

    def __init__(self, eng):
        if not getattr(eng, "lifted", False):
            raise ValueError("the engine handle was not created with detect_simple_bounds")
        self._eng = eng
        self.n, self.m, self.nnz, self.m_full = eng.n, eng.m, eng.nnz, eng.m_full
        _, var = eng.simple_rows()
        self.kept_row, _, _ = eng.lift_map(with_bounds=False)
        self.keep_rows = np.nonzero(self.kept_row >= 0)[0]      # full rows that stay in g, in the reduced problem's order
        self.lifted_rows = np.nonzero(self.kept_row < 0)[0]
        self.lifted_vars = var[self.lifted_rows].astype(np.int64)
        self._snapshot = None   # (params generation, bounds of the lifted rows, bounds of their variables)

    def __getattr__(self, name):   # bounds, sparsity, eval, eval_hess, cost_terms, row_blocks, set_params, ...: the engine's own
        return getattr(self._eng, name)

    def _lifted_bounds(self):
        """bounds of the lifted rows and of their variables for the engine's CURRENT parameters (taken again whenever set_params ran:
        a snapshot made by an earlier bounds() call would assign multipliers by the previous initial / final state)"""
        gen = getattr(self._eng, "params_generation", None)
        if self._snapshot is None or gen is None or self._snapshot[0] != gen:
            _, lb_full, ub_full = self._eng.lift_map(with_bounds=True)
            lbx, ubx, _, _ = self._eng.bounds()
            v = self.lifted_vars
            self._snapshot = (gen, lb_full[self.lifted_rows], ub_full[self.lifted_rows], lbx[v], ubx[v])
        return self._snapshot[1:]

    def full_multipliers(self, lam_reduced, lam_x=None):
        """lambda over ALL rows of the reference's subject_to list from the reduced problem's row multipliers and the multipliers
        lam_x of the variable bounds (signed, Lagrangian f + lambda^T g + lam_x^T x: IPOPT's mult_x_U - mult_x_L).  The multiplier of
        a variable's bound goes to the lifted row that defines the bound the variable ended up with (the first one, should several
        rows define the same bound); the other lifted rows of that variable are inactive."""
        lam = np.zeros(self.m_full)
        if lam_reduced is not None:
            lam[self.keep_rows] = np.asarray(lam_reduced, float).reshape(-1)
        if lam_x is not None:
            lx = np.asarray(lam_x, float).reshape(-1)
            v = self.lifted_vars
            lb_row, ub_row, lb_var, ub_var = self._lifted_bounds()
            defines = np.where(lx[v] >= 0.0, ub_row <= ub_var, lb_row >= lb_var)
            first = np.zeros(v.size, bool)
            seen = set()
            for i in np.nonzero(defines)[0]:
                if int(v[i]) not in seen:
                    seen.add(int(v[i]))
                    first[i] = True
            lam[self.lifted_rows] = np.where(first, lx[v], 0.0)
        return lam


def _hessian_evaluator(eng):
    """eng.eval_hess with ONE value array for the whole solve where the engine takes one (`out=`): the handle registers an output array
    it sees twice in a row and stores into it directly, and a fresh 1.2 MB numpy array per call would pay its page faults each time.
    The callers copy the values on (into a sparse matrix / into IPOPT's array) before the next evaluation."""
    import inspect
    try:
        takes_out = "out" in inspect.signature(eng.eval_hess).parameters
    except (TypeError, ValueError):
        takes_out = False
    keep = {}

    def evaluate(x, obj_factor, lam):
        if not takes_out:
            return np.asarray(eng.eval_hess(x, obj_factor, lam)).reshape(-1)
        keep["out"] = eng.eval_hess(x, obj_factor, lam, out=keep.get("out"))
        return keep["out"].reshape(-1)
    return evaluate


class _CallbackCache:
    """What IPOPT's `new_x` flag does for a C caller: the four callbacks of one iterate share ONE evaluation.

    On an engine handle (HipNlp, directly or behind the detect_simple_bounds view) this is the library's own host path, used the
    way INTEGRATION.md tells a C binding to use it: ONE set of output arrays for the whole solve (the handle registers them at
    their second sight: the kernel stores straight into them, no staging copy), early outputs on (the first callback at a new x
    fills all four arrays: one transfer per iterate), `new_x` unknown (the drivers without IPOPT's flag — cyipopt, SciPy — pass
    none: the library compares x with its staging copy instead of Python comparing and copying 151 KB per callback).  The callbacks
    hand out VIEWS of those arrays, valid until the next evaluation (cyipopt copies them
    into IPOPT's arrays at once; the SciPy driver asks for copies where it keeps a result).
    On anything else (the emulated engine of the CPU tests, the pose finder's 81-variable handle) it compares x itself."""

    def __init__(self, eng):
        self._eng = eng
        self._x = None
        self.calls = {}          # callbacks served, by kind
        self._evaluations = 0    # of which new evaluations (slow path; the fast path reads the handle's own counter)
        self._fast = all(hasattr(eng, name) for name in ("set_early_outputs", "host_stats", "register_outputs"))
        if self._fast:
            self._out = (np.empty(1), np.empty((1, eng.n)), np.empty((1, eng.m)), np.empty((1, eng.nnz)))
            self._pick = {w: tuple(o if k in w else None for k, o in zip(("f", "grad", "g", "jac"), self._out))
                          for w in (("f",), ("grad",), ("g",), ("jac",), ("f", "grad", "g", "jac"))}
            # the raw C entry point with the arrays' addresses fixed once: a callback is then ONE foreign call (the generic
            # HipNlp.eval spends ~3 us per call converting its arguments, four times per iterate)
            import ctypes as C
            handle = eng._eng if isinstance(eng, _SimpleBoundsLift) else eng
            self._h, self._n = handle.h, handle.n
            self._call = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)(
                ("hipnlp_eval", handle.lib))
            self._addr = {w: tuple(None if o is None else o.ctypes.data for o in out) for w, out in self._pick.items()}
            self._check = handle._check
            eng.set_early_outputs(True, grad=True)    # (the arrays above are this object's own scratch: grad f may be early too)
            self._evaluations0 = eng.host_stats()["evaluations"]

    @property
    def evaluations(self):
        return self._eng.host_stats()["evaluations"] - self._evaluations0 if self._fast else self._evaluations

    def eval(self, x, want, copy=False):
        """A non-finite evaluation (HIPNLP_E_NUMERIC) is NOT an exception here: the arrays come back as the kernel filled them and
        the NLP driver sees the NaN / Inf, as IPOPT does with CasADi's (it cuts the step and goes on; a Python exception out of a
        cyipopt callback would abort the solve).  Every other engine error still raises."""
        kind = want[0] if len(want) == 1 else "all"
        self.calls[kind] = self.calls.get(kind, 0) + 1
        if self._fast:
            out = self._pick.get(want)
            if out is not None and isinstance(x, np.ndarray) and x.dtype == np.float64 and x.size == self._n and x.flags.c_contiguous:
                rc = self._call(self._h, x.ctypes.data, -1, *self._addr[want])     # new_x unknown: the library compares
                if rc != 0 and rc != -5:   # (-5: non-finite values, handed on as they are)
                    self._check(rc)
            else:
                out = self._eng.eval(x, new_x=None, want=want, out=out or tuple(o if k in want else None for k, o in zip(("f", "grad", "g", "jac"), self._out)),
                                     nan_ok=True)
            return tuple(None if o is None else o.copy() for o in out) if copy else out
        x = np.asarray(x, dtype=np.float64)
        new_x = self._x is None or not np.array_equal(x, self._x)
        if new_x:
            self._x = x.copy()
        self._evaluations += int(new_x)
        try:
            return self._eng.eval(x[None, :], new_x=new_x, want=want, nan_ok=True)
        except Exception:
            self._x = None   # a failed evaluation leaves nothing to reuse
            raise

    def invalidate(self):
        self._x = None

    def close(self):
        """the solve is over: the registered arrays go away with this object"""
        if self._fast:
            self._eng.set_early_outputs(False)
            self._eng.unregister_outputs(self._out[1:])


class HipNlpSolver(OptimizationSolver):
    def __init__(self, settings, model, device=0, inner_solver="auto", options_solver=None, problem="kinodynamic",
                 callback_criterion: CallbackCriterion = None, callback_save_costs=True, callback_save_constraint_multipliers=True,
                 error_on_fail=True, detect_simple_bounds=True, jac_varying_first=True, pin_to_device_numa_node=False, devices=None):
        """callback_* as in OptiSolver (opti_solver.py:105-131).  error_on_fail: CasADi's Opti raises when IPOPT does not report
        success (e.g. Maximum_Iterations_Exceeded), which is what triggers the best-iterate fallback of opti_solver.py:479-520;
        False keeps the last iterate of an unconverged run instead (useful for smoke runs with a few iterations).
        jac_varying_first: the order in which the engine hands jac g to the NLP driver as triplets — True (default): the varying
        entries of a knot block first (the host path then moves them alone); False: CasADi's CCS order, what the reference's nlpsol
        hands IPOPT (opti_solver.py:479).  The drivers build their matrix from the triplets, so the order cannot change a solve:
        tests/test_gpu_solver_order.py runs the planners through both and compares every iterate.
        pin_to_device_numa_node: restrict the calling thread to the CPUs of the card's NUMA node before the engine and its arrays are
        created (hippopt_amd.hipnlp.pin_to_device_numa_node: the callbacks are link-bound, 2 - 5 us per call slower from the other
        socket of the host).  Off by default: it changes the affinity of the CALLER's thread; a deployment does it once for the process.
        devices: HIP ordinals — the kinodynamic horizon is cut into one contiguous knot range per entry and every callback of the ONE NLP
        driver in this process is evaluated by all of them (hipnlp_multi_create: x staged once, every shard storing its entries straight
        into the driver's arrays over its own link); an ordinal may repeat.  None: one handle on `device`.
        iterate_trace (attribute): set to a list to have (iteration, x, cost, primal infeasibility) of every iterate appended."""
        self._callback_criterion = callback_criterion
        self._callback_save_costs = callback_save_costs
        self._callback_save_constraint_multipliers = callback_save_constraint_multipliers
        self._callback = None
        self._error_on_fail = error_on_fail
        # casadi_opti_options["detect_simple_bounds"] of the reference scripts (main_periodic_step.py:110): kinodynamic problem only
        self._detect_simple_bounds = bool(detect_simple_bounds) and problem == "kinodynamic"
        self._jac_varying_first = bool(jac_varying_first)
        self._pin = bool(pin_to_device_numa_node)
        self.iterate_trace = None
        if problem not in ("kinodynamic", "pose"):
            raise ValueError("problem must be 'kinodynamic' or 'pose'")
        self._problem_kind = problem
        self._settings, self._model, self._device = settings, model, device
        self._devices = None if devices is None else [int(d) for d in devices]
        if self._devices is not None and problem != "kinodynamic":
            raise ValueError("devices=[...] shards the kinodynamic horizon; the pose finder is a single knot")
        self._inner_solver = inner_solver
        self._options = dict(options_solver or {})
        self._logger = logging.getLogger("[hippopt_amd::HipNlpSolver]")
        self._structure = self._objects = self._guess = None
        self._var_index, self._par_index = {}, {}
        self._n = self._np = 0
        self._engine = None
        self._lift = None
        self._values = self._cost_value = None
        self._cost_values, self._multipliers = {}, {}
        self._problem = None

    # ---- structure -------------------------------------------------------------------------------------
    def generate_optimization_objects(self, input_structure, **kwargs):
        if not isinstance(input_structure, (OptimizationObject, list)):
            raise ValueError("The input structure is neither an optimization object nor a list.")
        self._structure = copy.deepcopy(input_structure)
        expanded = extend_structure_to_horizon(input_structure, **kwargs)
        values, meta = expanded.to_dicts()
        seen = {}
        for name, value in values.items():  # duplicate-object guard (opti_solver.py:275-300)
            if isinstance(value, np.ndarray):
                if id(value) in seen:
                    raise ValueError(f"{seen[id(value)]} and {name} share the same object as value.")
                seen[id(value)] = name
        xo = po = 0
        self._var_index, self._par_index = {}, {}
        for name, value in values.items():
            if value is None:
                raise ValueError("Field " + name + " is tagged as storage, but it is None.")
            if not isinstance(value, np.ndarray) or value.ndim != 2:
                raise ValueError(f"Field {name} is tagged as storage, but it is not a 2-D array.")
            if value.size == 0:
                raise ValueError("Field " + name + " has a zero dimension.")
            if meta[name][STORAGE_TYPE] == "variable":
                self._var_index[name] = (xo, value.size, value.shape)
                xo += value.size
            elif meta[name][STORAGE_TYPE] == "parameter":
                self._par_index[name] = (po, value.size, value.shape)
                po += value.size
            else:
                raise ValueError("Unsupported input storage type")
        self._n, self._np = xo, po
        horizon = int(kwargs.get("horizon", 1))
        if self._problem_kind == "pose":
            if xo != _abi.POSE_NX or po != _abi.POSE_NP:
                raise ValueError(f"the structure is not the pose finder Variables tree the engine evaluates (variables {xo}, parameters {po})")
        elif xo != _abi.NXK * horizon + _abi.NXG or po != _abi.NPK * horizon + _abi.NPG:
            raise ValueError("the structure is not the kinodynamic Variables tree the engine evaluates "
                             f"(variables {xo}, parameters {po}, horizon {horizon})")
        self._objects = expanded
        if kwargs.get("fill_initial_guess", True):
            try:
                self.set_initial_guess(expanded)
            except Exception as err:  # noqa: BLE001
                raise InitialGuessFailure(err)
        return self._objects

    def get_optimization_objects(self):
        return self._objects

    def get_optimization_structure(self):
        return self._structure

    def register_problem(self, problem):
        self._problem = problem

    def get_problem(self):
        return self._problem

    # ---- guesses / parameters -----------------------------------------------------------------------------
    def set_initial_guess(self, initial_guess):
        """Variables -> x0, parameters -> p.  `None` leaves keep their previous value (opti_solver.py:379-420)."""
        if self._guess is None:
            self._guess = copy.deepcopy(self._objects)
        flat = initial_guess.to_dict()
        current = self._guess.to_dict()
        update = {}
        for name, value in flat.items():
            if value is None or name not in current:
                continue
            arr = np.asarray(value, dtype=float)
            target = self._var_index.get(name) or self._par_index.get(name)
            if target is None:
                continue
            if arr.size != target[1]:
                raise ValueError(f"The guess for {name} has {arr.size} entries, expected {target[1]}")
            update[name] = arr.reshape(target[2])
        self._guess.from_dict(update)

    def get_initial_guess(self):
        return copy.deepcopy(self._guess)

    def _pack(self):
        flat = self._guess.to_dict()
        x, p = np.zeros(self._n), np.zeros(self._np)
        for name, (off, size, _) in self._var_index.items():
            x[off:off + size] = np.asarray(flat[name], float).reshape(-1)
        for name, (off, size, _) in self._par_index.items():
            p[off:off + size] = np.asarray(flat[name], float).reshape(-1)
        return x, p

    # ---- engine -------------------------------------------------------------------------------------------
    def engine(self):
        if self._engine is None:
            from .hipnlp import HipNlp, HipPose, pin_to_device_numa_node  # raises loudly without the library / a device
            if self._pin:
                pin_to_device_numa_node(self._devices[0] if self._devices else self._device)
            if self._problem_kind == "pose":
                self._engine = _PoseEngine(HipPose(self._settings, self._model, batch=1, device=self._device))
            else:
                # (the NLP drivers take jac g as triplets — IPOPT's jacobianstructure / a COO matrix: inside a knot's block the entries
                #  that depend on x come first, so the host path stores ONE contiguous run per knot and leaves the constant ones alone)
                self._engine = HipNlp(self._settings, self._model, batch=1, device=self._device, detect_simple_bounds=self._detect_simple_bounds,
                                      jac_varying_first=self._jac_varying_first, devices=self._devices)
        return self._engine

    def get_constraint_expressions(self):
        """{constraint base name: (first row, rows per knot, first knot, knots)} — the typed counterpart of the MX dictionary."""
        return {b[0]: b[1:] for b in self.engine().row_blocks()}

    def get_cost_expressions(self):
        names, _ = self.engine().cost_terms() if self._values is not None else (None, None)
        return names

    def add_cost(self, input_cost, name=None):
        raise TypedProblemError("add_cost")

    def add_constraint(self, input_constraint, name=None):
        raise TypedProblemError("add_constraint")

    def cost_function(self):
        return self.get_cost_expressions()

    # ---- solve --------------------------------------------------------------------------------------------
    def nlp_view(self):
        """the engine as the NLP driver sees it.  With detect_simple_bounds the handle is the reduced problem (single-variable rows
        are bounds) and the view adds the way back from its multipliers to every named constraint."""
        eng = self.engine()
        if getattr(eng, "lifted", False):
            if self._lift is None or self._lift._eng is not eng:
                self._lift = _SimpleBoundsLift(eng)
            return self._lift
        return eng

    def solve(self):
        full = self.engine()
        x0, p = self._pack()
        full.set_params(p[None, :])
        eng = self.nlp_view()
        lbx, ubx, lbg, ubg = eng.bounds()
        ir, jc = eng.sparsity()

        solver = self._inner_solver
        if solver == "auto":
            try:
                import cyipopt  # noqa: F401
                solver = "ipopt"
            except ImportError:
                solver = "trust-constr"
        use_callback = self._callback_criterion is not None
        self._callback = None
        if use_callback:   # opti_solver.py:451-477
            self._callback = SaveBestUnsolvedVariablesCallback(self._callback_criterion, self._callback_save_costs,
                                                               self._callback_save_constraint_multipliers)

        def cost_values_at(xk):
            # through the SAME cache as the callbacks: the engine's cached result then always belongs to cache._x (an evaluation
            # behind the cache's back would let a later new_x = False callback at the older point read this point's outputs)
            self._cache.eval(xk, ("f",))
            names, terms = eng.cost_terms()
            return {n: float(v) for n, v in zip(names, terms[0])}
        self._cost_values_at = cost_values_at
        self._cache = _CallbackCache(eng)
        failure = None
        try:
            if solver == "ipopt":
                x, lam, info = self._solve_ipopt(eng, x0, lbx, ubx, lbg, ubg, ir, jc)
            else:
                x, lam, info = self._solve_scipy(eng, x0, lbx, ubx, lbg, ubg, ir, jc)
            lam = self._expand_multipliers(eng, lam, info.get("lam_x"))
            info["callbacks"] = dict(self._cache.calls, evaluations=self._cache.evaluations)
            info["nlp"] = {"n": eng.n, "m": eng.m, "nnz": int(len(ir)), "simple_bounds_lifted": int(getattr(full, "m_full", full.m) - eng.m)}
            if self._error_on_fail and not info.get("success", True):
                failure = RuntimeError("solver status: " + str(info.get("message", info.get("status"))))
        except Exception as err:  # noqa: BLE001
            failure, info = err, {"success": False, "message": str(err)}
        finally:
            self._cache.close()
        self._last_info = info
        if failure is not None:   # opti_solver.py:479-520: fall back to the best iterate the callback saved, else raise
            cb = self._callback
            if not use_callback or cb.best_iteration is None:
                raise HipFailure(failure, callback_used=use_callback)
            self._logger.warning("The solver failed to solve the problem, but the callback managed to save an intermediate "
                                 f"solution at iteration {cb.best_iteration}.")
            x = cb.best_x
            lam = cb.best_constraint_multipliers if cb.best_constraint_multipliers is not None else np.zeros(getattr(full, "m_full", full.m))
            self._cost_value = float(cb.best_cost)
            self._cost_values = dict(cb.best_cost_values) if self._callback_save_costs else {}
            self._store_solution(full, x, lam, with_multipliers=self._callback_save_constraint_multipliers)
            return
        f, *_ = full.eval(x[None, :], want=("f",))
        self._cost_value = float(f[0])
        names, terms = full.cost_terms()
        self._cost_values = {n: float(v) for n, v in zip(names, terms[0])}
        self._store_solution(full, x, lam)

    @staticmethod
    def _expand_multipliers(eng, lam, lam_x=None):
        """multipliers over every row of the engine's layout (the named constraints of Output.constraint_multipliers), whether or not
        the NLP driver saw the reduced problem"""
        if isinstance(eng, _SimpleBoundsLift) and lam is not None:
            return eng.full_multipliers(lam, lam_x)
        return lam

    def _store_solution(self, eng, x, lam, with_multipliers=True):
        self._multipliers = {}
        if with_multipliers:
            for name, first, rows, k0, nk in eng.row_blocks():
                self._multipliers[name] = np.asarray(lam[first:first + rows * nk]).reshape(nk, rows)
        values = copy.deepcopy(self._guess)
        update = {name: x[off:off + size].reshape(shape) for name, (off, size, shape) in self._var_index.items()}
        values.from_dict(update)
        self._values = values

    def _iterate_callback(self, iteration, x, cost, inf_pr, multipliers, lam_x=None):
        if self.iterate_trace is not None:
            self.iterate_trace.append((int(iteration), np.array(x, dtype=float, copy=True), float(cost), float(inf_pr)))
        if self._callback is not None:
            if multipliers is not None and self._lift is not None and getattr(self.engine(), "lifted", False):
                multipliers = self._lift.full_multipliers(multipliers, lam_x)
            self._callback(IterateInfo(int(iteration), float(cost), float(inf_pr)), x, multipliers,
                           (lambda: self._cost_values_at(np.asarray(x, float))) if self._callback_save_costs else None)

    def _solve_scipy(self, eng, x0, lbx, ubx, lbg, ubg, ir, jc):
        from scipy.optimize import BFGS, Bounds, NonlinearConstraint, minimize
        from scipy.sparse import csc_matrix
        m, n = eng.m, eng.n

        cache = self._cache

        def fun(x):
            f, *_ = cache.eval(x, ("f",))
            return float(f[0])

        def grad(x):   # (copies: this driver keeps gradients and constraint values of earlier points)
            _, g_, *_ = cache.eval(x, ("grad",), copy=True)
            return g_[0]

        def cons(x):
            _, _, g, _ = cache.eval(x, ("g",), copy=True)
            return g[0]

        def jac(x):
            _, _, _, j = cache.eval(x, ("jac",))
            return csc_matrix((j[0], (ir, jc)), shape=(m, n))
        exact = hasattr(eng, "eval_hess") and self._options.get("hessian_approximation", "exact") != "limited-memory"
        hess_f, hess_c = BFGS(), BFGS()
        if exact:
            try:
                hr, hc = eng.hess_sparsity()
            except HipNlpError as err:   # a part the engine reports as not built -> quasi-Newton, as with `limited-memory`
                if err.code != -6:
                    raise
                exact = False
        if exact:   # exact second derivatives from the engine (lower triangle -> symmetric matrix)
            off = hr != hc

            def sym(vals):
                return csc_matrix((np.concatenate([vals, vals[off]]), (np.concatenate([hr, hc[off]]), np.concatenate([hc, hr[off]]))), shape=(n, n))
            zero_lam = np.zeros(m)
            hessian = _hessian_evaluator(eng)

            def hess_f(x):   # (the Hessian evaluation reuses the engine's staging of x: the cached callback set is gone)
                cache.invalidate()
                return sym(hessian(x[None, :], 1.0, zero_lam[None, :]))

            def hess_c(x, v):
                cache.invalidate()
                return sym(hessian(x[None, :], 0.0, np.asarray(v)[None, :]))
        nlc = NonlinearConstraint(cons, lbg, ubg, jac=jac, hess=hess_c)
        opts = {"maxiter": int(self._options.get("max_iter", 50)), "verbose": int(self._options.get("verbose", 0)),
                "gtol": float(self._options.get("tol", 1e-6))}
        bounded = bool(np.any(np.isfinite(lbx)) or np.any(np.isfinite(ubx)))   # (lifted single-variable rows: detect_simple_bounds)
        # IPOPT's termination tests on the stand-in driver (the reference scripts stop on them: tol 1e-3, constr_viol_tol 1e-4,
        # acceptable_tol 10, acceptable_iter 2, acceptable_obj_change_tol 1, main_single_step_flat_ground.py:105-130): optimal when
        # max(|grad L|_inf, constraint violation) <= tol and the violation <= constr_viol_tol; "acceptable" when the same error is
        # <= acceptable_tol, the violation <= acceptable_constr_viol_tol and the relative cost change <= acceptable_obj_change_tol
        # for acceptable_iter consecutive iterations.  (IPOPT scales its error by the multiplier norms; this one is unscaled.)
        o = self._options
        tol, cv_tol = float(o.get("tol", 1e-8)), float(o.get("constr_viol_tol", 1e-4))
        acc_tol, acc_iter = float(o.get("acceptable_tol", 1e-6)), int(o.get("acceptable_iter", 15))
        acc_cv, acc_obj = float(o.get("acceptable_constr_viol_tol", 1e-2)), float(o.get("acceptable_obj_change_tol", 1e20))
        stop = {"reason": None, "run": 0, "f_prev": None}
        ipopt_tests = any(k in o for k in ("constr_viol_tol", "acceptable_tol", "acceptable_iter"))   # only when the script sets them

        def on_iterate(xk, state):
            self._iterate_callback(state.nit, xk, state.fun, state.constr_violation, state.v[0] if len(state.v) else None,
                                   state.v[1] if bounded and len(state.v) > 1 else None)
            if not ipopt_tests:
                return False
            err = max(float(state.optimality), float(state.constr_violation))
            if err <= tol and state.constr_violation <= cv_tol:
                stop["reason"] = "Solve_Succeeded"
                return True
            df = abs(state.fun - stop["f_prev"]) / max(1.0, abs(state.fun)) if stop["f_prev"] is not None else np.inf
            stop["f_prev"] = float(state.fun)
            stop["run"] = stop["run"] + 1 if (err <= acc_tol and state.constr_violation <= acc_cv and df <= acc_obj) else 0
            if acc_iter > 0 and stop["run"] >= acc_iter:
                stop["reason"] = "Solved_To_Acceptable_Level"
                return True
            return False
        res = minimize(fun, x0, jac=grad, hess=hess_f, constraints=[nlc], bounds=Bounds(lbx, ubx) if bounded else None,
                       method="trust-constr", options=opts, callback=on_iterate)
        lam = res.v[0] if len(res.v) else np.zeros(m)
        return res.x, lam, {"status": stop["reason"] or res.status, "success": stop["reason"] is not None or res.status in (1, 2),
                            "message": stop["reason"] or res.message, "iterations": res.nit, "optimality": float(res.optimality),
                            "constr_violation": res.constr_violation, "lam_x": res.v[1] if bounded and len(res.v) > 1 else None}

    def _solve_ipopt(self, eng, x0, lbx, ubx, lbg, ubg, ir, jc):
        import cyipopt
        outer = self

        cache = self._cache

        class Callbacks:
            def objective(self, x):
                return float(cache.eval(x, ("f",))[0][0])

            def gradient(self, x):
                return cache.eval(x, ("grad",))[1][0]

            def constraints(self, x):
                return cache.eval(x, ("g",))[2][0]

            def jacobianstructure(self):
                return ir, jc

            def jacobian(self, x):
                self.x_last = np.array(x, copy=True)
                return cache.eval(x, ("jac",))[3][0]

            def intermediate(self, alg_mod, iter_count, obj_value, inf_pr, inf_du, mu, d_norm, regularization_size, alpha_du, alpha_pr,
                             ls_trials):
                x_it = getattr(self, "x_last", x0)   # cyipopt < 1.3 has no get_current_iterate: the last point the Jacobian saw
                lam_it = lam_x_it = None
                try:
                    it = nlp.get_current_iterate()
                    x_it, lam_it = it["x"], it["mult_g"]
                    lam_x_it = np.asarray(it["mult_x_U"]) - np.asarray(it["mult_x_L"])
                except Exception:  # noqa: BLE001
                    pass
                outer._iterate_callback(iter_count, x_it, obj_value, inf_pr, lam_it, lam_x_it)
                return True
        exact = hasattr(eng, "eval_hess") and outer._options.get("hessian_approximation", "exact") != "limited-memory"
        if exact:
            try:
                hr, hc = eng.hess_sparsity()
            except HipNlpError as err:   # a part the engine reports as not built -> limited-memory
                if err.code != -6:
                    raise
                exact = False
        if exact:   # eval_h from the engine: the pose finder runs IPOPT with the exact Hessian (humanoid_pose_finder/main.py:101)
            Callbacks.hessianstructure = lambda self: (hr, hc)
            hessian = _hessian_evaluator(eng)

            def _hessian(self, x, lagrange, obj_factor):
                cache.invalidate()   # (the Hessian evaluation reuses the engine's staging of x)
                return hessian(x[None, :], obj_factor, np.asarray(lagrange)[None, :])
            Callbacks.hessian = _hessian
        nlp = cyipopt.Problem(n=eng.n, m=eng.m, problem_obj=Callbacks(), lb=lbx, ub=ubx, cl=lbg, cu=ubg)
        if not exact:
            nlp.add_option("hessian_approximation", "limited-memory")  # main_periodic_step.py:116
        for k, v in outer._options.items():
            if k in ("verbose", "detect_simple_bounds"):   # options of the stand-in driver / of Opti, not of IPOPT
                continue
            nlp.add_option(k, v)
        x, info = nlp.solve(x0)
        info = dict(info)
        info["success"] = info.get("status", -1) in (0, 1)   # Solve_Succeeded / Solved_To_Acceptable_Level
        info["message"] = info.get("status_msg", "")
        if "mult_x_U" in info and "mult_x_L" in info:
            info["lam_x"] = np.asarray(info["mult_x_U"]) - np.asarray(info["mult_x_L"])
        return x, info["mult_g"], info

    def get_values(self):
        return self._values

    def get_cost_value(self):
        return self._cost_value

    def get_cost_values(self):
        return self._cost_values

    def get_constraint_multipliers(self):
        return self._multipliers
