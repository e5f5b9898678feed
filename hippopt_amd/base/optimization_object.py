"""Dataclass trees of optimisation variables / parameters and their flat {name: array} form.

Host-side mirror of hippopt's `OptimizationObject` contract (reference: base/optimization_object.py:20-351,
base/variable.py:15-53, base/parameter.py:10-44) without CasADi: same field metadata vocabulary, same flat
names ("a.b", "list[2].c"), same ordering (dataclass field order, lists in index order), same override rule
(an `Overridable*` leaf takes the storage type of the closest composite ancestor that declares one).
"""
import dataclasses
from enum import Enum
from typing import Any, Callable, ClassVar, TypeVar

import numpy as np

TOptimizationObject = TypeVar("TOptimizationObject", bound="OptimizationObject")
StorageType = Any
CompositeType = Any

STORAGE_TYPE, TIME_DEPENDENT, TIME_EXPANSION, OVERRIDE_IF_COMPOSITE, COMPOSITE_TYPE = (
    "StorageType", "TimeDependent", "TimeExpansion", "OverrideIfComposite", "CompositeType")


class TimeExpansion(Enum):
    List = 0
    Matrix = 1


def _to_2d(value):
    """Numbers and 1-D data become column arrays; everything else is returned untouched (optimization_object.py:43-62)."""
    if isinstance(value, list) and (len(value) == 0 or all(isinstance(e, (int, float)) and not isinstance(e, bool) for e in value)):
        value = np.array(value, dtype=float)
    if isinstance(value, np.ndarray):
        return value.reshape(-1, 1) if value.ndim < 2 else value
    if isinstance(value, (int, float)) and not isinstance(value, bool):
        return float(value) * np.ones((1, 1))
    return value


@dataclasses.dataclass
class OptimizationObject:
    StorageTypeValue: ClassVar[str] = "generic"
    StorageTypeField: ClassVar[str] = STORAGE_TYPE
    TimeDependentField: ClassVar[str] = TIME_DEPENDENT
    TimeExpansionField: ClassVar[str] = TIME_EXPANSION
    OverrideIfCompositeField: ClassVar[str] = OVERRIDE_IF_COMPOSITE
    CompositeTypeField: ClassVar[str] = COMPOSITE_TYPE
    StorageTypeMetadata: ClassVar[dict] = {STORAGE_TYPE: "generic", TIME_DEPENDENT: False, TIME_EXPANSION: TimeExpansion.List,
                                           OVERRIDE_IF_COMPOSITE: False}
    IsValueFilter: ClassVar[Callable] = staticmethod(lambda _name, value, _meta: isinstance(value, np.ndarray))

    @classmethod
    def default_storage_metadata(cls, **kwargs) -> dict:
        return {}

    # ---- traversal ----------------------------------------------------------------------------------
    @staticmethod
    def _is_object_list(value):
        return isinstance(value, list) and len(value) > 0 and all(isinstance(e, (OptimizationObject, list)) for e in value)

    @staticmethod
    def _walk(obj, prefix, inherited, visit):
        """Depth-first traversal in declaration order.  visit(owner, field, index or None, full_name, value, metadata)."""
        if isinstance(obj, list):   # (a list of lists — objects expanded in time inside a list — is named a[i][k].leaf)
            for i, elem in enumerate(obj):
                OptimizationObject._walk(elem, f"{prefix}[{i}]" + ("" if isinstance(elem, list) else "."), inherited, visit)
            return
        for field in dataclasses.fields(obj):
            value = getattr(obj, field.name)
            if isinstance(value, OptimizationObject) or OptimizationObject._is_object_list(value):
                meta = field.metadata.get(COMPOSITE_TYPE)
                child_inherited = inherited
                if meta is not None and not (inherited is not None and meta.get(OVERRIDE_IF_COMPOSITE, False)):
                    child_inherited = meta
                sep = "" if isinstance(value, list) else "."
                OptimizationObject._walk(value, prefix + field.name + sep, child_inherited, visit)
                continue
            if STORAGE_TYPE not in field.metadata:
                continue
            meta = dict(field.metadata)
            if meta.get(OVERRIDE_IF_COMPOSITE, False) and inherited is not None and STORAGE_TYPE in inherited:
                meta[STORAGE_TYPE] = inherited[STORAGE_TYPE]
            as_array = _to_2d(value)
            if isinstance(as_array, list):
                for i in range(len(value)):
                    visit(obj, field, i, f"{prefix}{field.name}[{i}]", value[i], meta)
            else:
                visit(obj, field, None, prefix + field.name, value, meta)

    @staticmethod
    def _nest(flat: dict, prefix: str = "") -> dict:
        """{"a.b[2].c": v} -> {"a": {"b": [.., .., {"c": v}]}}; a prefix becomes one more outer level."""
        nested: dict = {}
        for name, value in flat.items():
            node = nested
            parts = name.split(".")
            for depth, part in enumerate(parts):
                key, idx = (part[:part.index("[")], int(part[part.index("[") + 1:-1])) if part.endswith("]") else (part, None)
                last = depth == len(parts) - 1
                if idx is None:
                    if last:
                        node[key] = value
                    else:
                        node = node.setdefault(key, {})
                else:
                    lst = node.setdefault(key, [])
                    while len(lst) <= idx:
                        lst.append(None if last else {})
                    if last:
                        lst[idx] = value
                    else:
                        node = lst[idx]
        return {prefix: nested} if prefix else nested

    def to_dicts(self, prefix: str = "", output_filter: Callable | None = None, output_conversion: Callable | None = None,
                 flatten: bool = True):
        values, metadata = {}, {}

        def visit(_, __, ___, name, value, meta):
            if output_conversion is not None:
                value = output_conversion(name, value)
            value = _to_2d(value)
            if output_filter is not None and not output_filter(name, value, meta):
                return
            values[name] = value
            metadata[name] = meta
        OptimizationObject._walk(self, prefix if flatten else "", None, visit)
        if flatten:
            return values, metadata
        return OptimizationObject._nest(values, prefix), OptimizationObject._nest(metadata, prefix)

    def to_dict(self, prefix: str = "", output_filter=None, output_conversion=None, flatten: bool = True) -> dict:
        return self.to_dicts(prefix=prefix, output_filter=output_filter, output_conversion=output_conversion, flatten=flatten)[0]

    def from_dict(self, input_dict: dict, prefix: str = "", input_conversion: Callable | None = None) -> None:
        def visit(owner, field, index, name, _value, _meta):
            if name not in input_dict:
                return
            new = input_dict[name]
            if input_conversion is not None:
                new = input_conversion(name, new)
            if index is None:
                setattr(owner, field.name, new)
            else:
                getattr(owner, field.name)[index] = new
        OptimizationObject._walk(self, prefix, None, visit)

    def to_list(self, output_filter=None, output_conversion=None) -> list:
        flat = self.to_dict(output_filter=output_filter, output_conversion=output_conversion)
        return [flat[k] for k in sorted(flat.keys())]   # sorted flat keys (optimization_object.py:300-312)


def default_storage_metadata(cls, **kwargs) -> dict:
    return cls.default_storage_metadata(**kwargs)


def default_storage_field(cls, **kwargs):
    return dataclasses.field(default=None, metadata=default_storage_metadata(cls, **kwargs))


def time_varying_metadata(time_varying: bool = True):
    return {TIME_DEPENDENT: time_varying}


def default_composite_field(cls=None, factory=None, time_varying: bool = True):
    meta = time_varying_metadata(time_varying)
    meta[COMPOSITE_TYPE] = cls.StorageTypeMetadata if cls is not None else None
    return dataclasses.field(default_factory=factory, metadata=meta)


def _storage_class(name, override):
    @dataclasses.dataclass
    class _Storage(OptimizationObject):
        StorageTypeValue: ClassVar[str] = name
        StorageTypeMetadata: ClassVar[dict] = {STORAGE_TYPE: name, TIME_DEPENDENT: name == "variable" or override,
                                               TIME_EXPANSION: TimeExpansion.List, OVERRIDE_IF_COMPOSITE: override}

        @classmethod
        def default_storage_metadata(cls, time_dependent: bool = None, time_expansion: TimeExpansion = TimeExpansion.List, **_) -> dict:
            meta = dict(cls.StorageTypeMetadata)
            if time_dependent is not None:
                meta[TIME_DEPENDENT] = time_dependent
            meta[TIME_EXPANSION] = time_expansion
            return meta
    return _Storage


# base/variable.py:15-53 and base/parameter.py:10-44: variables are time dependent by default, parameters are not;
# the Overridable* flavours take the storage type of a composite parent that declares one
Variable = _storage_class("variable", False)
Variable.__name__ = "Variable"
Variable.StorageTypeMetadata = {STORAGE_TYPE: "variable", TIME_DEPENDENT: True, TIME_EXPANSION: TimeExpansion.List, OVERRIDE_IF_COMPOSITE: False}
Parameter = _storage_class("parameter", False)
Parameter.__name__ = "Parameter"
Parameter.StorageTypeMetadata = {STORAGE_TYPE: "parameter", TIME_DEPENDENT: False, TIME_EXPANSION: TimeExpansion.List, OVERRIDE_IF_COMPOSITE: False}
OverridableVariable = _storage_class("variable", True)
OverridableVariable.__name__ = "OverridableVariable"
OverridableVariable.StorageTypeMetadata = {STORAGE_TYPE: "variable", TIME_DEPENDENT: True, TIME_EXPANSION: TimeExpansion.List, OVERRIDE_IF_COMPOSITE: True}
OverridableParameter = _storage_class("parameter", True)
OverridableParameter.__name__ = "OverridableParameter"
OverridableParameter.StorageTypeMetadata = {STORAGE_TYPE: "parameter", TIME_DEPENDENT: False, TIME_EXPANSION: TimeExpansion.List, OVERRIDE_IF_COMPOSITE: True}
