"""Table-driven declaration of OptimizationObject dataclasses.

The reference spells every node of its variable trees out as a hand-written dataclass; here a node is ONE call of `declare`:
leaf arrays with their storage kind and default value, child nodes with their factory, constructor-only arguments, and an optional
`setup(self, **constructor_arguments)` hook.  What the flat layout depends on — field names, field order (with several bases:
reverse MRO, as for any dataclass) and the metadata `OptimizationObject` scans — is exactly what a hand-written class would give;
`tests/golden/kinodyn_structure.json` (produced by the reference's own classes) pins it.
"""
import dataclasses

from .optimization_object import OptimizationObject, default_composite_field, default_storage_field, time_varying_metadata


def leaf(kind, default=None):
    """a storage leaf of `kind` (Variable, Parameter, Overridable*): `default()` fills it when the constructor leaves it None"""
    return ("leaf", kind, default)


def child(factory, time_varying=True, kind=None):
    """a composite child built by `factory`; `kind` (Variable / Parameter): the storage kind its Overridable* leaves take"""
    return ("child", factory, (time_varying, kind))


def series(default=None):
    """a list of nodes that varies over time (no factory: the constructor or `setup` fills it), e.g. the masses of a toy OCP"""
    return ("series", default, None)


def plain(default=None):
    """an ordinary attribute (not scanned)"""
    return ("plain", default, None)


def argument(default=None):
    """a constructor-only argument, handed to `setup`"""
    return ("argument", default, None)


def declare(name, fields, bases=(OptimizationObject,), setup=None, methods=None, module=None):
    """fields: {name: leaf(...) | child(...) | series(...) | plain(...) | argument(...)} in layout order"""
    specs, defaults, arguments = [], {}, []
    for fname, (what, a, b) in fields.items():
        if what == "leaf":
            specs.append((fname, object, default_storage_field(a)))
            if b is not None:
                defaults[fname] = b
        elif what == "child":
            specs.append((fname, object, default_composite_field(cls=b[1], factory=a, time_varying=b[0])))
        elif what == "series":
            specs.append((fname, object, dataclasses.field(default=a, metadata=time_varying_metadata())))
        elif what == "plain":
            specs.append((fname, object, dataclasses.field(default=a)))
        else:
            specs.append((fname, dataclasses.InitVar[object], dataclasses.field(default=a)))
            arguments.append(fname)

    own = {"arguments": (), "defaults": {}}   # filled in below, once the class exists

    def post_init(self, *values, **named):
        """constructor-only arguments arrive positionally from the dataclass machinery (declaration order, bases first) or by name
        from a hand-written subclass that calls this node's __post_init__ itself"""
        given = dict(zip(own["arguments"], values))
        given.update(named)
        if setup is not None:
            setup(self, **given)
        for fname, make in own["defaults"].items():
            if getattr(self, fname) is None:
                setattr(self, fname, make())

    namespace = dict(methods or {})
    namespace["__post_init__"] = post_init
    cls = dataclasses.make_dataclass(name, specs, bases=tuple(bases), namespace=namespace)
    cls.__leaf_defaults__ = defaults
    for klass in reversed(cls.__mro__):   # defaults of the declared bases, then this node's own
        own["defaults"].update(klass.__dict__.get("__leaf_defaults__", {}))
    # every constructor-only argument the dataclass machinery passes to __post_init__, in its order (field order: those of the bases
    # first); the names are the ones the `declare` tables of this node and of its bases listed as argument()
    cls.__declared_arguments__ = tuple(arguments)
    declared = {n for klass in cls.__mro__ for n in klass.__dict__.get("__declared_arguments__", ())}
    own["arguments"] = tuple(n for n in cls.__dataclass_fields__ if n in declared)
    if module is not None:
        cls.__module__ = module
    return cls
