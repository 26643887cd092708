"""cwipc_util_amd -- MI355X-native implementation of cwipc_util's per-point filter path.

Drop-in for the part of ``import cwipc`` that the filter path uses (reference
python/cwipc/__init__.py does ``from .util import *`` as well):

    import cwipc_util_amd as cwipc
"""
from .util import *   # noqa: F401,F403
from . import util as util   # noqa: F401
