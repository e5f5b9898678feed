"""Horizon expansion of an optimisation structure (transcription rules T1/T2 of SURVEY §8a).

Mirror of MultipleShootingSolver._extend_structure_to_horizon (reference: base/multiple_shooting_solver.py:64-181) and
of the flattened naming of `_generate_flattened_and_symbolic_objects` (:220-493): every time-dependent top-level field is
replaced by a list of `horizon` deep copies (TimeExpansion.List) or by a matrix whose columns are the knots
(TimeExpansion.Matrix); fields flagged `time_varying=False` stay single.  `horizons={field: n}` overrides per field.
"""
import copy
import dataclasses

import numpy as np

from .optimization_object import (COMPOSITE_TYPE, STORAGE_TYPE, TIME_DEPENDENT, TIME_EXPANSION, OptimizationObject, TimeExpansion)


def extend_structure_to_horizon(input_structure, **kwargs):
    if "horizon" not in kwargs and "horizons" not in kwargs:
        return input_structure
    default_len = int(kwargs.get("horizon", 1))
    if default_len < 1:
        raise ValueError("The specified horizon needs to be a strictly positive integer")
    horizons = kwargs.get("horizons") if isinstance(kwargs.get("horizons"), dict) else {}
    out = copy.deepcopy(input_structure)
    for field in dataclasses.fields(out):
        n = default_len
        constant = TIME_DEPENDENT in field.metadata and not field.metadata[TIME_DEPENDENT]
        custom = field.name in horizons
        if custom:
            constant = False
            n = int(horizons[field.name])
            if n < 1:
                raise ValueError("The specified horizon for " + field.name + " needs to be a strictly positive integer")
        if constant:
            continue
        value = getattr(out, field.name)
        if STORAGE_TYPE in field.metadata:
            if field.metadata.get(TIME_EXPANSION) is TimeExpansion.Matrix:
                if not isinstance(value, np.ndarray):
                    raise ValueError("Field " + field.name + " is not a Numpy array. Cannot expand it to the horizon.")
                if value.ndim > 1 and value.shape[1] > 1:
                    raise ValueError("Cannot expand " + field.name + " since it is already a matrix.")
                col = value.reshape(-1, 1) if value.ndim < 2 else value
                setattr(out, field.name, np.tile(col, (1, n)))
            else:
                setattr(out, field.name, [copy.deepcopy(value) for _ in range(n)])
            continue
        if TIME_DEPENDENT not in field.metadata and not custom:
            continue  # nested objects are expanded only when flagged time dependent (or given a custom horizon)
        if isinstance(value, OptimizationObject):
            setattr(out, field.name, [copy.deepcopy(value) for _ in range(n)])
        elif isinstance(value, list) and len(value) and all(isinstance(e, OptimizationObject) for e in value):
            setattr(out, field.name, [[copy.deepcopy(e) for _ in range(n)] for e in value])
    return out


def flattened_names(expanded, original):
    """{flat name without the time index: (horizon length, [flat names per knot])}.  In the reference the time list is NOT
    part of the flattened name ("system.contact_points.left[0].p", multiple_shooting_solver.py:293-485)."""
    out = {}
    for field in dataclasses.fields(expanded):
        value, orig = getattr(expanded, field.name), getattr(original, field.name)
        expanded_in_time = isinstance(value, list) and not isinstance(orig, list) and \
            (STORAGE_TYPE in field.metadata or isinstance(orig, OptimizationObject))
        if expanded_in_time:
            per_knot = []
            for k, elem in enumerate(value):
                if isinstance(elem, OptimizationObject):
                    per_knot.append(list(elem.to_dict(prefix=f"{field.name}[{k}].").keys()))
                else:
                    per_knot.append([f"{field.name}[{k}]"])
            base = [n.replace(f"{field.name}[0]", field.name, 1) for n in per_knot[0]]
            for j, b in enumerate(base):
                out[b] = (len(value), [names[j] for names in per_knot])
        elif STORAGE_TYPE in field.metadata:
            out[field.name] = (value.shape[1] if field.metadata.get(TIME_EXPANSION) is TimeExpansion.Matrix and isinstance(value, np.ndarray) else 1,
                               [field.name])
        elif isinstance(value, OptimizationObject):
            for n in value.to_dict(prefix=field.name + ".").keys():
                out[n] = (1, [n])
        elif isinstance(value, list) and len(value) and all(isinstance(e, list) for e in value):  # list of objects, each expanded
            for i, series in enumerate(value):
                per_knot = [list(e.to_dict(prefix=f"{field.name}[{i}][{k}].").keys()) for k, e in enumerate(series)]
                for j, n0 in enumerate(per_knot[0]):
                    out[n0.replace(f"{field.name}[{i}][0]", f"{field.name}[{i}]", 1)] = (len(series), [names[j] for names in per_knot])
        elif isinstance(value, list) and len(value) and all(isinstance(e, OptimizationObject) for e in value):
            for i, e in enumerate(value):
                for n in e.to_dict(prefix=f"{field.name}[{i}].").keys():
                    out[n] = (1, [n])
    return out
