"""Small object-graph helpers with the reference's names (open_flamingo/src/utils.py:1-48)."""
from functools import reduce


def extend_instance(obj, mixin):
    """Re-type ``obj`` in place so that ``mixin`` precedes its original class in the MRO (reference utils.py:1-7):
    the mixin's ``forward`` then runs first and reaches the original one through ``super()``."""
    base = obj.__class__
    obj.__class__ = type(base.__name__, (mixin, base), {})


def getattr_recursive(obj, att):
    """``getattr_recursive(o, 'a.b.c') == o.a.b.c``; the empty path returns ``o`` (reference utils.py:10-21)."""
    return reduce(getattr, att.split("."), obj) if att else obj


def setattr_recursive(obj, att, val):
    """``setattr_recursive(o, 'a.b.c', v)`` sets ``o.a.b.c = v`` (reference utils.py:24-31)."""
    head, _, leaf = att.rpartition(".")
    setattr(getattr_recursive(obj, head), leaf, val)


def apply_with_stopping_condition(module, apply_fn, apply_condition=None, stopping_condition=None, **other_args):
    """Pre-order walk that prunes sub-trees where ``stopping_condition`` holds (reference utils.py:34-48)."""
    stack = [module]
    while stack:
        m = stack.pop()
        if stopping_condition(m):
            continue
        if apply_condition(m):
            apply_fn(m, **other_args)
        stack.extend(reversed(list(m.children())))
