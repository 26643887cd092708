"""Filter plugins of the hot path (reference python/cwipc/filters/__init__.py:19-48).

Only the filters on the MI355X path exist here: voxelize, remove_outliers, crop,
colorize, transform, simulatecams (plus passthrough).  The factory accepts the reference's FILTERDESC
syntax -- "name" or "name(args)" -- but parses the arguments with
ast.literal_eval instead of eval.
"""
import ast
from typing import cast

from .abstract import cwipc_abstract_filter
from . import passthrough, voxelize, crop, remove_outliers, colorize, transform, simulatecams

all_filters = [passthrough, voxelize, crop, remove_outliers, colorize, transform, simulatecams]
_by_name = {m.CustomFilter.filtername: m for m in all_filters}


def factory(filterdesc: str) -> cwipc_abstract_filter:
    """Create a filter from a description such as 'voxelize(0.01)' or 'passthrough'."""
    if filterdesc.endswith(')'):
        openpos = filterdesc.find('(')
        name = filterdesc[:openpos]
        args = ast.literal_eval(filterdesc[openpos:])
        if type(args) != type(()):
            args = (args,)
    else:
        name, args = filterdesc, ()
    if name not in _by_name:
        raise ValueError(f"unknown filter {name!r}; available: {sorted(_by_name)}")
    return cast(cwipc_abstract_filter, _by_name[name].CustomFilter(*args))
