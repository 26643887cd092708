"""ctypes wrapper around the MI355X libcwipc_util -- same surface as the reference's
``cwipc.util`` (reference python/cwipc/util.py), so pipeline code written against
``cwipc`` runs against ``cwipc_util_amd`` unchanged:

    import cwipc_util_amd as cwipc
    pc = cwipc.cwipc_synthetic().get()
    small = cwipc.cwipc_downsample(pc, 0.01)

Same names, argument meaning and error behaviour (CwipcError where the reference
raises it, wrapper objects around whatever the C call returns where it does not).
The per-point work happens in HIP kernels behind the C-ABI; this module never
computes on points itself and has no CPU fallback.  open3d is optional here (the
reference imports it unconditionally, util.py:24).
"""
from __future__ import annotations

import ctypes
import functools
import os
import sys
import warnings
from typing import Any, Callable, Dict, Iterable, List, Optional, Tuple, Union

import numpy
import numpy.typing

from .abstract import cwipc_pointcloud_abstract, cwipc_source_abstract, cwipc_activesource_abstract, cwipc_tileinfo_dict

__all__ = [
    'CWIPC_API_VERSION', 'CWIPC_POINT_PACKETHEADER_MAGIC', 'CWIPC_POINT_PACKETHEADER_MAGIC_C', 'cwipc_proxy_packet', 'cwipc_from_proxy_packet', 'CWIPC_FLAGS_BINARY', 'CwipcError',
    'CWIPC_LOG_LEVEL_NONE', 'CWIPC_LOG_LEVEL_ERROR', 'CWIPC_LOG_LEVEL_WARNING', 'CWIPC_LOG_LEVEL_TRACE', 'CWIPC_LOG_LEVEL_DEBUG',
    'cwipc_pointcloud_wrapper', 'cwipc_source_wrapper', 'cwipc_activesource_wrapper', 'cwipc_sink_wrapper', 'cwipc_metadata',
    'cwipc_point', 'cwipc_point_array', 'cwipc_point_numpy_dtype', 'cwipc_tileinfo_dict', 'cwipc_point_packetheader',
    'cwipc_util_dll_load',
    'cwipc_get_version', 'cwipc_log_configure', 'cwipc_log_default_callback', '_cwipc_log_emit', 'cwipc_dangling_allocations',
    'cwipc_read', 'cwipc_read_debugdump', 'cwipc_write', 'cwipc_write_debugdump',
    'cwipc_from_points', 'cwipc_from_packet', 'cwipc_from_numpy_array', 'cwipc_from_numpy_matrix', 'cwipc_from_o3d_pointcloud',
    'cwipc_synthetic', 'cwipc_capturer', 'cwipc_window', 'cwipc_proxy',
    'cwipc_downsample', 'cwipc_remove_outliers', 'cwipc_tilefilter', 'cwipc_tilemap', 'cwipc_colormap',
    'cwipc_join', 'cwipc_join_multi', 'cwipc_crop',
    # MI355X extensions (no reference counterpart)
    'cwipc_hip_device_count', 'cwipc_hip_set_device', 'cwipc_hip_upload', 'cwipc_hip_pinned_points', 'cwipc_hip_pin_array', 'cwipc_hip_colorize', 'cwipc_tilefilter_masked', 'cwipc_hip_device_planes',
    'cwipc_hip_profile', 'cwipc_hip_knn_mean_dist', 'cwipc_hip_from_device_aos', 'cwipc_hip_from_device_slots', 'cwipc_hip_copy_device_aos',
    'cwipc_transform', 'cwipc_offset_scale', 'get_tiles_used', 'cwipc_downsample_pertile', 'cwipc_hip_simulatecams', 'cwipc_hip_comm', 'cwipc_hip_comm_unique_id',
]

# reference util.py:86, 346, 348
CWIPC_API_VERSION = 0x20260129
CWIPC_POINT_PACKETHEADER_MAGIC = 0x20210208
CWIPC_FLAGS_BINARY = 1

# reference util.py:356-361
CWIPC_LOG_LEVEL_NONE = 0
CWIPC_LOG_LEVEL_ERROR = 1
CWIPC_LOG_LEVEL_WARNING = 2
CWIPC_LOG_LEVEL_TRACE = 3
CWIPC_LOG_LEVEL_DEBUG = 4


class CwipcError(RuntimeError):
    pass


# ---------------------------------------------------------------------------
# native handle and record types (reference util.py:236-340)
# ---------------------------------------------------------------------------
class cwipc_pointcloud_p(ctypes.c_void_p):
    """native pointer to a cwipc_pointcloud"""


class cwipc_source_p(ctypes.c_void_p):
    """native pointer to a cwipc_source"""


class cwipc_activesource_p(cwipc_source_p):
    """native pointer to a cwipc_activesource"""


class cwipc_sink_p(ctypes.c_void_p):
    """native pointer to a cwipc_sink"""


class cwipc_metadata_p(ctypes.c_void_p):
    """native pointer to a cwipc_metadata"""


class _FieldwiseEq(ctypes.Structure):
    def __eq__(self, other: Any) -> bool:
        return isinstance(other, type(self)) and all(getattr(self, f[0]) == getattr(other, f[0]) for f in self._fields_)

    def __ne__(self, other: Any) -> bool:
        return not self.__eq__(other)

    __hash__ = None  # type: ignore


class cwipc_point(_FieldwiseEq):
    """One point: x, y, z (float), r, g, b (0..255), tile (8 bit number or camera mask).  16 bytes."""
    _fields_ = [("x", ctypes.c_float), ("y", ctypes.c_float), ("z", ctypes.c_float),
                ("r", ctypes.c_ubyte), ("g", ctypes.c_ubyte), ("b", ctypes.c_ubyte), ("tile", ctypes.c_ubyte)]


cwipc_point_numpy_dtype = [('x', '<f4'), ('y', '<f4'), ('z', '<f4'), ('r', 'u1'), ('g', 'u1'), ('b', 'u1'), ('tile', 'u1')]
assert ctypes.sizeof(cwipc_point) == 16 and numpy.dtype(cwipc_point_numpy_dtype).itemsize == 16


class cwipc_vector(_FieldwiseEq):
    _fields_ = [("x", ctypes.c_double), ("y", ctypes.c_double), ("z", ctypes.c_double)]


class cwipc_tileinfo(ctypes.Structure):
    _fields_ = [("normal", cwipc_vector), ("cameraName", ctypes.c_char_p), ("ncamera", ctypes.c_uint8), ("cameraMask", ctypes.c_uint8)]


class cwipc_point_packetheader(ctypes.Structure):
    """Header for talking to a cwipc_proxy server (field order as the reference's Python side, util.py:335-344)."""
    _fields_ = [("hdr", ctypes.c_uint32), ("magic", ctypes.c_uint32), ("cellsize", ctypes.c_float),
                ("timestamp", ctypes.c_uint64), ("unused", ctypes.c_uint32), ("dataCount", ctypes.c_uint32)]


cwipc_log_callback_type = Callable[[int, bytes], None]
_cwipc_log_callback_t = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_char_p)
_cwipc_log_callback_ref = None

# ---------------------------------------------------------------------------
# library loading (reference util.py:368-553)
# ---------------------------------------------------------------------------
_dll: Optional[ctypes.CDLL] = None

_c = ctypes
_ERR = _c.POINTER(_c.c_char_p)
_BYTES = _c.POINTER(_c.c_byte)

# name -> (argtypes, restype); every symbol the reference binds, then the extensions
_SIGNATURES: Dict[str, Tuple[list, Any]] = {
    'cwipc_get_version': ([], _c.c_char_p),
    'cwipc_log_configure': ([_c.c_int, _cwipc_log_callback_t], None),
    'cwipc_dangling_allocations': ([_c.c_bool], _c.c_int),
    '_cwipc_log_emit': ([_c.c_int, _c.c_char_p, _c.c_char_p], None),
    'cwipc_read': ([_c.c_char_p, _c.c_ulonglong, _ERR, _c.c_ulong], cwipc_pointcloud_p),
    'cwipc_write_ext': ([_c.c_char_p, cwipc_pointcloud_p, _c.c_int, _ERR], _c.c_int),
    'cwipc_from_points': ([_c.c_void_p, _c.c_size_t, _c.c_int, _c.c_ulonglong, _ERR, _c.c_ulong], cwipc_pointcloud_p),
    'cwipc_from_packet': ([_c.c_char_p, _c.c_size_t, _ERR, _c.c_ulong], cwipc_pointcloud_p),
    'cwipc_read_debugdump': ([_c.c_char_p, _ERR, _c.c_ulong], cwipc_pointcloud_p),
    'cwipc_write_debugdump': ([_c.c_char_p, cwipc_pointcloud_p, _ERR], _c.c_int),
    'cwipc_pointcloud_free': ([cwipc_pointcloud_p], None),
    'cwipc_pointcloud__shallowcopy': ([cwipc_pointcloud_p], cwipc_pointcloud_p),
    'cwipc_pointcloud_timestamp': ([cwipc_pointcloud_p], _c.c_ulonglong),
    'cwipc_pointcloud_cellsize': ([cwipc_pointcloud_p], _c.c_float),
    'cwipc_pointcloud__set_cellsize': ([cwipc_pointcloud_p, _c.c_float], None),
    'cwipc_pointcloud__set_timestamp': ([cwipc_pointcloud_p, _c.c_ulonglong], None),
    'cwipc_pointcloud_count': ([cwipc_pointcloud_p], _c.c_int),
    'cwipc_pointcloud_get_uncompressed_size': ([cwipc_pointcloud_p], _c.c_size_t),
    'cwipc_pointcloud_copy_uncompressed': ([cwipc_pointcloud_p, _BYTES, _c.c_size_t], _c.c_int),
    'cwipc_pointcloud_copy_packet': ([cwipc_pointcloud_p, _BYTES, _c.c_size_t], _c.c_size_t),
    'cwipc_pointcloud_access_metadata': ([cwipc_pointcloud_p], cwipc_metadata_p),
    'cwipc_activesource_start': ([cwipc_source_p], _c.c_bool),
    'cwipc_activesource_stop': ([cwipc_source_p], None),
    'cwipc_source_get': ([cwipc_source_p], cwipc_pointcloud_p),
    'cwipc_source_available': ([cwipc_source_p, _c.c_bool], _c.c_bool),
    'cwipc_source_eof': ([cwipc_source_p], _c.c_bool),
    'cwipc_source_free': ([cwipc_source_p], None),
    'cwipc_activesource_request_metadata': ([cwipc_source_p, _c.c_char_p], None),
    'cwipc_activesource_is_metadata_requested': ([cwipc_source_p, _c.c_char_p], _c.c_bool),
    'cwipc_activesource_reload_config': ([cwipc_activesource_p, _c.c_char_p], _c.c_bool),
    'cwipc_activesource_get_config': ([cwipc_activesource_p, _BYTES, _c.c_size_t], _c.c_size_t),
    'cwipc_activesource_seek': ([cwipc_activesource_p, _c.c_uint64], _c.c_bool),
    'cwipc_activesource_maxtile': ([cwipc_activesource_p], _c.c_int),
    'cwipc_activesource_get_tileinfo': ([cwipc_activesource_p, _c.c_int, _c.POINTER(cwipc_tileinfo)], _c.c_int),
    'cwipc_activesource_auxiliary_operation': ([cwipc_source_p, _c.c_char_p, _BYTES, _c.c_size_t, _BYTES, _c.c_size_t], _c.c_bool),
    'cwipc_sink_free': ([cwipc_sink_p], None),
    'cwipc_sink_feed': ([cwipc_sink_p, cwipc_pointcloud_p, _c.c_bool], _c.c_bool),
    'cwipc_sink_caption': ([cwipc_sink_p, _c.c_char_p], _c.c_bool),
    'cwipc_sink_interact': ([cwipc_sink_p, _c.c_char_p, _c.c_char_p, _c.c_int32], _c.c_char),
    'cwipc_synthetic': ([_c.c_int, _c.c_int, _ERR, _c.c_ulong], cwipc_activesource_p),
    'cwipc_capturer': ([_c.c_char_p, _ERR, _c.c_ulong], cwipc_activesource_p),
    'cwipc_window': ([_c.c_char_p, _ERR, _c.c_ulong], cwipc_sink_p),
    'cwipc_downsample': ([cwipc_pointcloud_p, _c.c_float], cwipc_pointcloud_p),
    'cwipc_remove_outliers': ([cwipc_pointcloud_p, _c.c_int, _c.c_float, _c.c_bool], cwipc_pointcloud_p),
    'cwipc_tilefilter': ([cwipc_pointcloud_p, _c.c_int], cwipc_pointcloud_p),
    'cwipc_tilemap': ([cwipc_pointcloud_p, _c.c_char_p], cwipc_pointcloud_p),
    'cwipc_colormap': ([cwipc_pointcloud_p, _c.c_ulong, _c.c_ulong], cwipc_pointcloud_p),
    'cwipc_crop': ([cwipc_pointcloud_p, _c.c_float * 6], cwipc_pointcloud_p),
    'cwipc_join': ([cwipc_pointcloud_p, cwipc_pointcloud_p], cwipc_pointcloud_p),
    'cwipc_proxy': ([_c.c_char_p, _c.c_int, _ERR, _c.c_ulong], cwipc_activesource_p),
    'cwipc_metadata_count': ([cwipc_metadata_p], _c.c_int),
    'cwipc_metadata_name': ([cwipc_metadata_p, _c.c_int], _c.c_char_p),
    'cwipc_metadata_description': ([cwipc_metadata_p, _c.c_int], _c.c_char_p),
    'cwipc_metadata_pointer': ([cwipc_metadata_p, _c.c_int], _c.c_void_p),
    'cwipc_metadata_size': ([cwipc_metadata_p, _c.c_int], _c.c_int),
    # ---- include/cwipc_util_amd/hip_ext.h ----
    'cwipc_hip_device_count': ([], _c.c_int),
    'cwipc_hip_set_device': ([_c.c_int], _c.c_int),
    'cwipc_hip_get_device': ([], _c.c_int),
    'cwipc_hip_last_error': ([], _c.c_char_p),
    'cwipc_hip_synchronize': ([], None),
    'cwipc_hip_pool_bytes': ([], _c.c_size_t),
    'cwipc_hip_pool_trim': ([], None),
    'cwipc_hip_workspace_trim': ([], _c.c_size_t),
    'cwipc_hip_host_alloc': ([_c.c_size_t], _c.c_void_p),
    'cwipc_hip_host_free': ([_c.c_void_p], None),
    'cwipc_hip_host_register': ([_c.c_void_p, _c.c_size_t], _c.c_int),
    'cwipc_hip_host_unregister': ([_c.c_void_p], _c.c_int),
    'cwipc_hip_upload': ([cwipc_pointcloud_p], _c.c_int),
    'cwipc_hip_drop_host_copy': ([cwipc_pointcloud_p], _c.c_int),
    'cwipc_hip_is_device_resident': ([cwipc_pointcloud_p], _c.c_int),
    'cwipc_hip_device_planes': ([cwipc_pointcloud_p, _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_void_p),
                                 _c.POINTER(_c.c_void_p), _c.POINTER(_c.c_size_t)], _c.c_int),
    'cwipc_hip_copy_device_aos': ([cwipc_pointcloud_p, _c.c_void_p, _c.c_size_t], _c.c_long),
    'cwipc_hip_from_device_aos': ([_c.c_void_p, _c.c_size_t, _c.c_uint64, _c.c_float], cwipc_pointcloud_p),
    'cwipc_hip_from_device_slots': ([_c.c_void_p, _c.c_int, _c.c_size_t, _c.c_size_t, _c.POINTER(_c.c_uint32), _c.c_uint64, _c.c_float], cwipc_pointcloud_p),
    'cwipc_hip_from_device_slots_on_stream': ([_c.c_void_p, _c.c_int, _c.c_size_t, _c.c_size_t, _c.POINTER(_c.c_uint32), _c.c_uint64, _c.c_float, _c.c_void_p], cwipc_pointcloud_p),
    'cwipc_hip_copy_device_aos_on_stream': ([cwipc_pointcloud_p, _c.c_void_p, _c.c_size_t, _c.c_void_p], _c.c_long),
    'cwipc_hip_colorize': ([cwipc_pointcloud_p, _c.c_double, _c.c_void_p, _c.c_void_p], cwipc_pointcloud_p),
    'cwipc_hip_join_multi': ([_c.POINTER(cwipc_pointcloud_p), _c.c_int], cwipc_pointcloud_p),
    'cwipc_hip_simulatecams': ([cwipc_pointcloud_p, _c.c_int, _c.c_float, _c.c_float, _c.c_void_p], cwipc_pointcloud_p),
    'cwipc_hip_tilefilter_masked': ([cwipc_pointcloud_p, _c.c_int], cwipc_pointcloud_p),
    'cwipc_hip_transform': ([cwipc_pointcloud_p, _c.POINTER(_c.c_double)], cwipc_pointcloud_p),
    'cwipc_hip_offset_scale': ([cwipc_pointcloud_p, _c.c_double, _c.c_double, _c.c_double, _c.c_double], cwipc_pointcloud_p),
    'cwipc_hip_tiles_used': ([cwipc_pointcloud_p, _c.POINTER(_c.c_ubyte)], _c.c_int),
    'cwipc_hip_knn_mean_dist': ([cwipc_pointcloud_p, _c.c_int, _c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_double), _c.c_float], _c.c_int),
    'cwipc_hip_workspace_bytes': ([], _c.c_size_t),
    'cwipc_hip_comm_unique_id': ([_c.c_void_p, _c.POINTER(_c.c_char_p)], _c.c_int),
    'cwipc_hip_comm_create': ([_c.c_void_p, _c.c_int, _c.c_int, _c.POINTER(_c.c_char_p)], _c.c_void_p),
    'cwipc_hip_comm_free': ([_c.c_void_p], None),
    'cwipc_hip_comm_rank': ([_c.c_void_p], _c.c_int),
    'cwipc_hip_comm_nranks': ([_c.c_void_p], _c.c_int),
    'cwipc_hip_comm_join': ([_c.c_void_p, cwipc_pointcloud_p, _c.c_int], cwipc_pointcloud_p),
    'cwipc_hip_comm_submit': ([_c.c_void_p, cwipc_pointcloud_p, _c.c_int], cwipc_pointcloud_p),
    'cwipc_hip_proxy_packet': ([cwipc_pointcloud_p, _c.c_void_p, _c.c_size_t, _c.c_uint32], _c.c_size_t),
    'cwipc_hip_from_proxy_packet': ([_c.c_void_p, _c.c_size_t, _c.c_int, _c.POINTER(_c.c_char_p), _c.c_uint64], cwipc_pointcloud_p),
    'cwipc_hip_profile_enable': ([_c.c_int], None),
    'cwipc_hip_profile_reset': ([], None),
    'cwipc_hip_profile_count': ([], _c.c_int),
    'cwipc_hip_profile_get': ([_c.c_int, _c.POINTER(_c.c_char_p), _c.POINTER(_c.c_double), _c.POINTER(_c.c_long)], _c.c_int),
}


def _default_library_path() -> Optional[str]:
    here = os.path.dirname(os.path.abspath(__file__))
    candidates = []
    if 'CWIPC_LIBRARY_DIR' in os.environ:   # same override the reference honours (util.py:120-123)
        candidates.append(os.path.join(os.environ['CWIPC_LIBRARY_DIR'], 'libcwipc_util.so'))
    candidates.append(os.path.join(here, 'lib', 'libcwipc_util.so'))
    for c in candidates:
        if os.path.exists(c):
            return c
    return None


def cwipc_util_dll_load(libname: Optional[str] = None) -> ctypes.CDLL:
    """Load libcwipc_util and declare the signatures (once).  Raises RuntimeError when the
    library is missing: there is no pure-Python or CPU fallback behind this wrapper."""
    global _dll
    if _dll is not None:
        return _dll
    path = libname if libname and os.path.isabs(libname) else _default_library_path()
    if not path:
        raise RuntimeError('Dynamic library cwipc_util not found (build it with `python -m cwipc_util_amd._build`)')
    dll = ctypes.CDLL(path)
    for name, (argtypes, restype) in _SIGNATURES.items():
        fn = getattr(dll, name)   # AttributeError here = the library does not export the boundary
        fn.argtypes = argtypes
        fn.restype = restype
    _dll = dll
    return dll


# ---------------------------------------------------------------------------
# point arrays
# ---------------------------------------------------------------------------
cwipc_point_array_value_type = Union[None, bytearray, bytes, ctypes.Array, List[tuple]]
cwipc_point_numpy_array_value_type = numpy.typing.NDArray[Any]
cwipc_point_numpy_matrix_value_type = numpy.typing.NDArray[numpy.floating]


def cwipc_point_array(*, count: Optional[int] = None, values: Any = ()) -> ctypes.Array:
    """Array of cwipc_point: `count` zeroed points, or built from a sequence of
    (x, y, z, r, g, b, tile) tuples, or wrapped around / copied from raw bytes."""
    if count is None:
        count = len(values)
    allocator = cwipc_point * count
    if isinstance(values, bytearray):
        return allocator.from_buffer(values)
    if isinstance(values, bytes):
        return allocator.from_buffer_copy(values)
    if not isinstance(values, tuple):
        values = tuple(values)
    return allocator(*values)


def _raise_or_warn(errorString: ctypes.c_char_p, rv: Any, fatal_if_no_rv: bool = True) -> None:
    if errorString and errorString.value:
        msg = errorString.value.decode('utf8')
        if not rv or not fatal_if_no_rv:
            raise CwipcError(msg)
        warnings.warn(msg)


# ---------------------------------------------------------------------------
# wrappers (reference util.py:573-1083)
# ---------------------------------------------------------------------------
class cwipc_pointcloud_wrapper(cwipc_pointcloud_abstract):
    """Point cloud as an opaque native object; freed when the wrapper is collected."""

    def __init__(self, _cwipc: Optional[cwipc_pointcloud_p] = None):
        if _cwipc is not None and not isinstance(_cwipc, cwipc_pointcloud_p):
            raise CwipcError("Invalid cwipc_pointcloud_p pointer passed to cwipc_pointcloud_wrapper")
        self._cwipc = _cwipc
        self._points = None
        self._bytes = None
        self._must_be_freed = True

    def __del__(self):
        if getattr(self, '_must_be_freed', False):
            self.free(force=True)

    def as_cwipc_p(self) -> cwipc_pointcloud_p:
        assert self._cwipc
        return self._cwipc

    def free(self, *, force: bool = False) -> None:
        if self._cwipc and self._must_be_freed:
            if not force:
                cwipc_log_default_callback(CWIPC_LOG_LEVEL_WARNING, b"cwipc_pointcloud_wrapper.free() called explicitly.")
            cwipc_util_dll_load().cwipc_pointcloud_free(self.as_cwipc_p())
        self._cwipc = None
        self._must_be_freed = False

    def detach(self) -> 'cwipc_pointcloud_wrapper':
        """Hand the native pointer to a new wrapper that will NOT free it; this wrapper becomes invalid."""
        if self._cwipc is None:
            cwipc_log_default_callback(CWIPC_LOG_LEVEL_WARNING, b"detach() called on NULL pointer")
        rv = type(self)(self._cwipc)
        rv._must_be_freed = False
        self._cwipc = None
        self._must_be_freed = False
        return rv

    def clone(self) -> 'cwipc_pointcloud_wrapper':
        return type(self)(cwipc_util_dll_load().cwipc_pointcloud__shallowcopy(self.as_cwipc_p()))

    def timestamp(self) -> int:
        return cwipc_util_dll_load().cwipc_pointcloud_timestamp(self.as_cwipc_p())

    def cellsize(self) -> float:
        return cwipc_util_dll_load().cwipc_pointcloud_cellsize(self.as_cwipc_p())

    def _set_cellsize(self, cellsize: float) -> None:
        cwipc_util_dll_load().cwipc_pointcloud__set_cellsize(self.as_cwipc_p(), cellsize)

    def _set_timestamp(self, timestamp: int) -> None:
        cwipc_util_dll_load().cwipc_pointcloud__set_timestamp(self.as_cwipc_p(), timestamp)

    def count(self) -> int:
        return cwipc_util_dll_load().cwipc_pointcloud_count(self.as_cwipc_p())

    def get_uncompressed_size(self) -> int:
        return cwipc_util_dll_load().cwipc_pointcloud_get_uncompressed_size(self.as_cwipc_p())

    def get_points(self) -> ctypes.Array:
        if self._points is None:
            self._initialize_points_and_bytes()
        return self._points

    def get_bytes(self) -> bytearray:
        if self._bytes is None:
            self._initialize_points_and_bytes()
        return self._bytes

    def get_numpy_array(self) -> cwipc_point_numpy_array_value_type:
        """Structured numpy view (fields x,y,z,r,g,b,tile) of the point data."""
        return numpy.ctypeslib.as_array(self.get_points())

    def get_numpy_matrix(self, onlyGeometry: bool = False) -> cwipc_point_numpy_matrix_value_type:
        """float32 matrix N x 7 (x, y, z, r, g, b, tile) or N x 3."""
        pts = self.get_numpy_array()
        m = numpy.zeros((pts.shape[0], 3 if onlyGeometry else 7), numpy.float32)
        m[:, 0], m[:, 1], m[:, 2] = pts['x'], pts['y'], pts['z']
        if not onlyGeometry:
            m[:, 3], m[:, 4], m[:, 5], m[:, 6] = pts['r'], pts['g'], pts['b'], pts['tile']
        return m

    def get_o3d_pointcloud(self):
        import open3d   # optional dependency
        m = self.get_numpy_matrix()
        pc = open3d.geometry.PointCloud()
        pc.points = open3d.utility.Vector3dVector(m[:, 0:3])
        pc.colors = open3d.utility.Vector3dVector(m[:, 3:6] / 255.0)
        return pc

    def access_metadata(self) -> Optional['cwipc_metadata']:
        rv_p = cwipc_util_dll_load().cwipc_pointcloud_access_metadata(self.as_cwipc_p())
        return cwipc_metadata(rv_p) if rv_p else None

    def _initialize_points_and_bytes(self) -> None:
        assert self._cwipc
        dll = cwipc_util_dll_load()
        nBytes = dll.cwipc_pointcloud_get_uncompressed_size(self.as_cwipc_p())
        buffer = bytearray(nBytes)
        bufferArg = (ctypes.c_byte * nBytes).from_buffer(buffer)
        nPoints = dll.cwipc_pointcloud_copy_uncompressed(self.as_cwipc_p(), bufferArg, nBytes)
        if nPoints < 0:
            raise CwipcError("cwipc_pointcloud_copy_uncompressed failed")
        self._points = cwipc_point_array(count=nPoints, values=buffer)
        self._bytes = buffer

    def copy_into(self, np_points: numpy.ndarray) -> int:
        """The points into a caller's contiguous numpy array of the cwipc_point dtype with exactly count() elements (the C call
        cwipc_pointcloud_copy_uncompressed, reference src/cwipc_util.cpp:226-250, without the bytearray get_points() makes): an
        array in page-locked memory (cwipc_hip_pinned_points, cwipc_hip_pin_array) is written by the GPU directly."""
        assert self._cwipc
        if not np_points.flags['C_CONTIGUOUS'] or np_points.dtype.itemsize != 16:
            raise ValueError("copy_into: a contiguous array of 16-byte cwipc_point records is needed")
        n = cwipc_util_dll_load().cwipc_pointcloud_copy_uncompressed(self.as_cwipc_p(), ctypes.cast(np_points.ctypes.data, _BYTES), np_points.nbytes)
        if n < 0:
            raise CwipcError("cwipc_pointcloud_copy_uncompressed failed")
        return n

    def get_packet(self) -> bytearray:
        assert self._cwipc
        dll = cwipc_util_dll_load()
        nBytes = dll.cwipc_pointcloud_copy_packet(self.as_cwipc_p(), None, 0)
        buffer = bytearray(nBytes)
        bufferArg = (ctypes.c_byte * nBytes).from_buffer(buffer)
        rvNBytes = dll.cwipc_pointcloud_copy_packet(self.as_cwipc_p(), bufferArg, nBytes)
        assert rvNBytes == nBytes
        return buffer


class cwipc_source_wrapper(cwipc_source_abstract):
    """Point cloud source as an opaque native object."""

    def __init__(self, _cwipc_source: Optional[cwipc_source_p] = None):
        if _cwipc_source is not None and not isinstance(_cwipc_source, cwipc_source_p):
            raise CwipcError("Invalid cwipc_source_p pointer passed to cwipc_source_wrapper")
        self._cwipc_source = _cwipc_source
        self._must_be_freed = True

    def __del__(self):
        if getattr(self, '_must_be_freed', False):
            self.free(force=True)

    def as_cwipc_source_p(self) -> cwipc_source_p:
        assert self._cwipc_source
        return self._cwipc_source

    def free(self, *, force: bool = False) -> None:
        if self._cwipc_source and self._must_be_freed:
            if not force:
                cwipc_log_default_callback(CWIPC_LOG_LEVEL_WARNING, b"cwipc_source_wrapper.free() called explicitly.")
            cwipc_util_dll_load().cwipc_source_free(self.as_cwipc_source_p())
        self._cwipc_source = None
        self._must_be_freed = False

    def detach(self) -> 'cwipc_source_wrapper':
        if self._cwipc_source is None:
            cwipc_log_default_callback(CWIPC_LOG_LEVEL_WARNING, b"detach() called on NULL pointer")
        rv = type(self)(self._cwipc_source)
        rv._must_be_freed = False
        self._cwipc_source = None
        self._must_be_freed = False
        return rv

    def eof(self) -> bool:
        return cwipc_util_dll_load().cwipc_source_eof(self.as_cwipc_source_p())

    def available(self, wait: bool) -> bool:
        return cwipc_util_dll_load().cwipc_source_available(self.as_cwipc_source_p(), wait)

    def get(self) -> Optional[cwipc_pointcloud_wrapper]:
        rv = cwipc_util_dll_load().cwipc_source_get(self.as_cwipc_source_p())
        return cwipc_pointcloud_wrapper(rv) if rv else None

    def statistics(self) -> None:
        pass


class cwipc_activesource_wrapper(cwipc_source_wrapper, cwipc_activesource_abstract):
    """Active (tiled) point cloud source as an opaque native object."""

    def __init__(self, _cwipc_activesource: Optional[cwipc_activesource_p] = None):
        if _cwipc_activesource is not None and not isinstance(_cwipc_activesource, cwipc_activesource_p):
            raise CwipcError("Invalid cwipc_activesource_p passed to cwipc_activesource_wrapper")
        cwipc_source_wrapper.__init__(self, _cwipc_activesource)

    def reload_config(self, config: Union[str, bytes, None]) -> bool:
        if isinstance(config, str):
            config = config.encode('utf8')
        return cwipc_util_dll_load().cwipc_activesource_reload_config(self.as_cwipc_source_p(), config)

    def get_config(self) -> bytes:
        dll = cwipc_util_dll_load()
        nBytes = dll.cwipc_activesource_get_config(self.as_cwipc_source_p(), None, 0)
        if nBytes <= 0:
            raise CwipcError("this cwipc_activesource has no camera configuration")
        buffer = bytearray(nBytes)
        bufferArg = (ctypes.c_byte * nBytes).from_buffer(buffer)
        assert dll.cwipc_activesource_get_config(self.as_cwipc_source_p(), bufferArg, nBytes) == nBytes
        return buffer

    def start(self) -> bool:
        return cwipc_util_dll_load().cwipc_activesource_start(self.as_cwipc_source_p())

    def stop(self) -> None:
        cwipc_util_dll_load().cwipc_activesource_stop(self.as_cwipc_source_p())

    def seek(self, timestamp: int) -> bool:
        return cwipc_util_dll_load().cwipc_activesource_seek(self.as_cwipc_source_p(), timestamp)

    def maxtile(self) -> int:
        return cwipc_util_dll_load().cwipc_activesource_maxtile(self.as_cwipc_source_p())

    def get_tileinfo_raw(self, tilenum: int) -> Optional[cwipc_tileinfo]:
        info = cwipc_tileinfo()
        rv = cwipc_util_dll_load().cwipc_activesource_get_tileinfo(self.as_cwipc_source_p(), tilenum, ctypes.byref(info))
        return info if rv else None

    def get_tileinfo_dict(self, tilenum: int) -> cwipc_tileinfo_dict:
        info = self.get_tileinfo_raw(tilenum)
        if info is None:
            raise CwipcError(f"get_tileinfo_raw({tilenum}) returned None")
        normal = dict(x=info.normal.x, y=info.normal.y, z=info.normal.z)
        return dict(normal=normal, cameraName=info.cameraName, ncamera=info.ncamera, cameraMask=info.cameraMask)

    def request_metadata(self, name: str) -> None:
        cwipc_util_dll_load().cwipc_activesource_request_metadata(self.as_cwipc_source_p(), name.encode('utf8'))

    def is_metadata_requested(self, name: str) -> bool:
        return cwipc_util_dll_load().cwipc_activesource_is_metadata_requested(self.as_cwipc_source_p(), name.encode('utf8'))

    def auxiliary_operation(self, op: str, inbuf: bytes, outbuf: bytearray) -> bool:
        c_inbuf = (ctypes.c_byte * len(inbuf)).from_buffer_copy(inbuf)
        c_outbuf = (ctypes.c_byte * len(outbuf)).from_buffer(outbuf)
        return cwipc_util_dll_load().cwipc_activesource_auxiliary_operation(
            self.as_cwipc_source_p(), op.encode('utf8'), c_inbuf, len(inbuf), c_outbuf, len(outbuf))


class cwipc_sink_wrapper:
    """Point cloud sink as an opaque native object (no sink is implemented by the MI355X build)."""

    def __init__(self, _cwipc_sink: Optional[cwipc_sink_p] = None):
        if _cwipc_sink is not None and not isinstance(_cwipc_sink, cwipc_sink_p):
            raise CwipcError("Invalid cwipc_sink_p passed to cwipc_sink_wrapper")
        self._cwipc_sink = _cwipc_sink
        self._must_be_freed = True

    def __del__(self):
        if getattr(self, '_must_be_freed', False):
            self.free(force=True)

    def as_cwipc_sink_p(self) -> cwipc_sink_p:
        assert self._cwipc_sink
        return self._cwipc_sink

    def free(self, *, force: bool = False) -> None:
        if self._cwipc_sink and self._must_be_freed:
            cwipc_util_dll_load().cwipc_sink_free(self.as_cwipc_sink_p())
        self._cwipc_sink = None
        self._must_be_freed = False

    def feed(self, pc: Optional[cwipc_pointcloud_wrapper], clear: bool) -> bool:
        return cwipc_util_dll_load().cwipc_sink_feed(self.as_cwipc_sink_p(), pc.as_cwipc_p() if pc is not None else None, clear)

    def caption(self, caption: str) -> bool:
        return cwipc_util_dll_load().cwipc_sink_caption(self.as_cwipc_sink_p(), caption.encode('utf8'))

    def interact(self, prompt: Optional[str], responses: Optional[str], millis: int) -> str:
        rv = cwipc_util_dll_load().cwipc_sink_interact(
            self.as_cwipc_sink_p(), prompt.encode('utf8') if prompt is not None else None,
            responses.encode('utf8') if responses is not None else None, millis)
        return rv.decode('utf8')


class cwipc_metadata:
    """Additional data attached to a point cloud (reference util.py:950-1083; image helpers omitted: capture is out of scope)."""

    def __init__(self, _cwipc_metadata: Optional[cwipc_metadata_p] = None):
        if _cwipc_metadata is not None:
            assert isinstance(_cwipc_metadata, cwipc_metadata_p)
        self._cwipc_metadata = _cwipc_metadata

    def as_cwipc_metadata_p(self) -> cwipc_metadata_p:
        assert self._cwipc_metadata
        return self._cwipc_metadata

    def count(self) -> int:
        return cwipc_util_dll_load().cwipc_metadata_count(self.as_cwipc_metadata_p())

    def name(self, idx: int) -> str:
        return cwipc_util_dll_load().cwipc_metadata_name(self.as_cwipc_metadata_p(), idx).decode('utf8')

    def description(self, idx: int) -> str:
        return cwipc_util_dll_load().cwipc_metadata_description(self.as_cwipc_metadata_p(), idx).decode('utf8')

    def pointer(self, idx: int) -> int:
        return cwipc_util_dll_load().cwipc_metadata_pointer(self.as_cwipc_metadata_p(), idx)

    def size(self, idx: int) -> int:
        return cwipc_util_dll_load().cwipc_metadata_size(self.as_cwipc_metadata_p(), idx)

    def data(self, idx: int) -> bytes:
        size = self.size(idx)
        return bytearray((ctypes.c_ubyte * size).from_address(self.pointer(idx)))


# ---------------------------------------------------------------------------
# module-level functions (reference util.py:1085-1343)
# ---------------------------------------------------------------------------
def cwipc_get_version() -> str:
    return cwipc_util_dll_load().cwipc_get_version().decode('utf8')


def cwipc_log_configure(level: int, callback: Optional[cwipc_log_callback_type] = None) -> None:
    global _cwipc_log_callback_ref
    _cwipc_log_callback_ref = _cwipc_log_callback_t(callback) if callback else _cwipc_log_callback_t(0)
    cwipc_util_dll_load().cwipc_log_configure(level, _cwipc_log_callback_ref)


def cwipc_log_default_callback(level: int, message: bytes) -> None:
    level_name = {1: "ERROR", 2: "WARNING", 3: "INFO", 4: "DEBUG"}.get(level, f"LEVEL{level}")
    print(f"{level_name}: cwipc: {message.decode('utf8')}", file=sys.stderr)


def _cwipc_log_emit(level: int, module: str, message: str) -> None:
    cwipc_util_dll_load()._cwipc_log_emit(level, module.encode('utf8'), message.encode('utf8'))


def cwipc_dangling_allocations(log: bool) -> int:
    return cwipc_util_dll_load().cwipc_dangling_allocations(log)


def cwipc_read(filename: str, timestamp: int) -> cwipc_pointcloud_wrapper:
    """Point cloud from a PLY file (ascii or binary; reference python/cwipc/util.py cwipc_read)."""
    errorString = ctypes.c_char_p()
    rv = cwipc_util_dll_load().cwipc_read(filename.encode('utf8'), timestamp, ctypes.byref(errorString), CWIPC_API_VERSION)
    _raise_or_warn(errorString, rv)
    if rv:
        return cwipc_pointcloud_wrapper(rv)
    raise CwipcError("cwipc_read: no pointcloud read, but no specific error returned from C library")


def cwipc_write(filename: str, pointcloud: cwipc_pointcloud_wrapper, flags: int = 0) -> int:
    errorString = ctypes.c_char_p()
    rv = cwipc_util_dll_load().cwipc_write_ext(filename.encode('utf8'), pointcloud.as_cwipc_p(), flags, ctypes.byref(errorString))
    _raise_or_warn(errorString, None)
    return rv


def cwipc_from_points(points: cwipc_point_array_value_type, timestamp: int) -> cwipc_pointcloud_wrapper:
    """Point cloud from a cwipc_point_array or a sequence of (x,y,z,r,g,b,tile) tuples."""
    if not isinstance(points, ctypes.Array):
        points = cwipc_point_array(values=points)
    errorString = ctypes.c_char_p()
    rv = cwipc_util_dll_load().cwipc_from_points(ctypes.addressof(points), ctypes.sizeof(points), len(points), timestamp,
                                                 ctypes.byref(errorString), CWIPC_API_VERSION)
    _raise_or_warn(errorString, None)
    if rv:
        return cwipc_pointcloud_wrapper(rv)
    raise CwipcError("cwipc_from_points: cannot create cwipc from given argument")


def cwipc_from_numpy_array(np_points: cwipc_point_numpy_array_value_type, timestamp: int) -> cwipc_pointcloud_wrapper:
    """Point cloud from a structured numpy array with the cwipc_point dtype."""
    nPoint = np_points.shape[0]
    np_points = numpy.ascontiguousarray(np_points)
    nBytes = nPoint * np_points.strides[0] if nPoint else 0
    errorString = ctypes.c_char_p()
    rv = cwipc_util_dll_load().cwipc_from_points(np_points.ctypes.data, nBytes, nPoint, timestamp, ctypes.byref(errorString), CWIPC_API_VERSION)
    _raise_or_warn(errorString, None)
    if rv:
        return cwipc_pointcloud_wrapper(rv)
    raise CwipcError("cwipc_from_numpy_array: cannot create cwipc from given argument")


def cwipc_from_numpy_matrix(np_points_matrix: cwipc_point_numpy_matrix_value_type, timestamp: int) -> cwipc_pointcloud_wrapper:
    """Point cloud from an N x 7 float matrix (x, y, z, r, g, b, tile)."""
    count = np_points_matrix.shape[0]
    assert np_points_matrix.shape == (count, 7)
    assert np_points_matrix.dtype in (numpy.float32, numpy.float64)
    np_points = numpy.zeros(count, cwipc_point_numpy_dtype)
    for col, name in enumerate(('x', 'y', 'z')):
        np_points[name] = np_points_matrix[:, col]
    for col, name in enumerate(('r', 'g', 'b', 'tile'), start=3):
        np_points[name] = np_points_matrix[:, col].astype(numpy.uint8)
    return cwipc_from_numpy_array(np_points, timestamp)


def cwipc_from_o3d_pointcloud(o3d_pc, timestamp: int) -> cwipc_pointcloud_wrapper:
    points = numpy.asarray(o3d_pc.points)
    colors = numpy.asarray(o3d_pc.colors)
    np_matrix = numpy.zeros((points.shape[0], 7))
    np_matrix[..., 0:3] = points
    np_matrix[..., 3:6] = colors * 256
    return cwipc_from_numpy_matrix(np_matrix, timestamp)


def cwipc_from_packet(packet: Union[bytes, bytearray]) -> cwipc_pointcloud_wrapper:
    nBytes = len(packet)
    byte_array_type = ctypes.c_char * nBytes
    try:
        c_packet = byte_array_type.from_buffer(packet)
    except TypeError:
        c_packet = byte_array_type.from_buffer_copy(packet)
    errorString = ctypes.c_char_p()
    rv = cwipc_util_dll_load().cwipc_from_packet(c_packet, nBytes, ctypes.byref(errorString), CWIPC_API_VERSION)
    _raise_or_warn(errorString, None)
    if rv:
        return cwipc_pointcloud_wrapper(rv)
    raise CwipcError("cwipc_from_packet: no pointcloud read, but no specific error returned from C library")


CWIPC_POINT_PACKETHEADER_MAGIC_C = 0x20201016   # what the reference's proxy SERVER checks (include/cwipc_util/api.h:110); the Python constant above is what its SENDER writes


def cwipc_proxy_packet(pc: cwipc_pointcloud_wrapper, magic: Optional[int] = None) -> bytes:
    """The bytes the reference's sender puts on the wire for `pc` (python/cwipc/scripts/cwipc_toproxy.py:51-57): a 24-byte
    cwipc_point_packetheader and the cwipc_point records.  magic: None = the C server's (0x20201016); the reference's Python
    sender writes CWIPC_POINT_PACKETHEADER_MAGIC (0x20210208), which its own server refuses."""
    dll = cwipc_util_dll_load()
    need = dll.cwipc_hip_proxy_packet(pc.as_cwipc_p(), None, 0, 0)
    if need == 0:
        raise CwipcError("cwipc_proxy_packet: NULL pointcloud")
    buf = bytearray(need)
    c_buf = (ctypes.c_char * need).from_buffer(buf)
    got = dll.cwipc_hip_proxy_packet(pc.as_cwipc_p(), ctypes.addressof(c_buf), need, 0 if magic is None else magic)
    del c_buf
    if got != need:
        raise CwipcError("cwipc_proxy_packet: could not build the packet")
    return bytes(buf)


def cwipc_from_proxy_packet(packet: Union[bytes, bytearray], accept_python_magic: bool = False) -> cwipc_pointcloud_wrapper:
    """What the reference's proxy server does with one packet (src/cwipc_proxy.cpp:179-216): the cloud, with the header's
    timestamp (the 8 bytes the server sends back) and cellsize."""
    n = len(packet)
    c_packet = (ctypes.c_char * n).from_buffer_copy(packet)
    errorString = ctypes.c_char_p()
    rv = cwipc_util_dll_load().cwipc_hip_from_proxy_packet(ctypes.addressof(c_packet), n, 1 if accept_python_magic else 0, ctypes.byref(errorString), CWIPC_API_VERSION)
    _raise_or_warn(errorString, None)
    if rv:
        return cwipc_pointcloud_wrapper(rv)
    raise CwipcError("cwipc_from_proxy_packet: no pointcloud read, but no specific error returned from C library")


def cwipc_read_debugdump(filename: str) -> cwipc_pointcloud_wrapper:
    errorString = ctypes.c_char_p()
    rv = cwipc_util_dll_load().cwipc_read_debugdump(filename.encode('utf8'), ctypes.byref(errorString), CWIPC_API_VERSION)
    _raise_or_warn(errorString, None)
    if rv:
        return cwipc_pointcloud_wrapper(rv)
    raise CwipcError("cwipc_read_debugdump: no pointcloud read, but no specific error returned from C library")


def cwipc_write_debugdump(filename: str, pointcloud: cwipc_pointcloud_wrapper) -> int:
    errorString = ctypes.c_char_p()
    rv = cwipc_util_dll_load().cwipc_write_debugdump(filename.encode('utf8'), pointcloud.as_cwipc_p(), ctypes.byref(errorString))
    _raise_or_warn(errorString, None)
    return rv


def cwipc_synthetic(fps: int = 0, npoints: int = 0) -> cwipc_activesource_wrapper:
    """Source producing synthetically generated point clouds on every get() call."""
    errorString = ctypes.c_char_p()
    rv = cwipc_util_dll_load().cwipc_synthetic(fps, npoints, ctypes.byref(errorString), CWIPC_API_VERSION)
    _raise_or_warn(errorString, None)
    if rv:
        return cwipc_activesource_wrapper(rv)
    raise CwipcError("cwipc_synthetic: cannot create synthetic source, but no specific error returned from C library")


def cwipc_capturer(conffile: Optional[str] = None) -> cwipc_activesource_wrapper:
    errorString = ctypes.c_char_p()
    rv = cwipc_util_dll_load().cwipc_capturer(conffile.encode('utf8') if conffile else None, ctypes.byref(errorString), CWIPC_API_VERSION)
    _raise_or_warn(errorString, rv)
    if rv:
        return cwipc_activesource_wrapper(rv)
    raise CwipcError("cwipc_capturer: cannot create capturer, but no specific error returned from C library")


def cwipc_window(title: str) -> cwipc_sink_wrapper:
    errorString = ctypes.c_char_p()
    rv = cwipc_util_dll_load().cwipc_window(title.encode('utf8'), ctypes.byref(errorString), CWIPC_API_VERSION)
    _raise_or_warn(errorString, None)
    if rv:
        return cwipc_sink_wrapper(rv)
    raise CwipcError("cwipc_window: cannot create window, but no specific error returned from C library")


def cwipc_proxy(host: str, port: int) -> cwipc_activesource_wrapper:
    errorString = ctypes.c_char_p()
    rv = cwipc_util_dll_load().cwipc_proxy(host.encode('utf8'), port, ctypes.byref(errorString), CWIPC_API_VERSION)
    _raise_or_warn(errorString, None)
    if rv:
        return cwipc_activesource_wrapper(rv)
    raise CwipcError("cwipc_proxy: cannot create capturer, but no specific error returned from C library")


# ---- the hot path (reference util.py:1284-1332) ----
# Unlike the reference thunks (which wrap whatever pointer comes back, NULL included),
# these raise CwipcError on NULL: on this build NULL usually means "no GPU", and a
# wrapper around NULL would only fail later and less legibly.
def _wrap_filter_result(name: str, rv: Any) -> cwipc_pointcloud_wrapper:
    if not rv:
        detail = cwipc_util_dll_load().cwipc_hip_last_error()
        raise CwipcError(f"{name}: C library returned NULL" + (f" ({detail.decode('utf8')})" if detail else ""))
    return cwipc_pointcloud_wrapper(rv)


def cwipc_downsample(pc: cwipc_pointcloud_wrapper, voxelsize: float) -> cwipc_pointcloud_wrapper:
    """Point cloud voxelized to cubes of the given size (negative: single pcl::VoxelGrid over the whole cloud)."""
    return _wrap_filter_result('cwipc_downsample', cwipc_util_dll_load().cwipc_downsample(pc.as_cwipc_p(), voxelsize))


def cwipc_remove_outliers(pc: cwipc_pointcloud_wrapper, kNeighbors: int, stdDesvMultThresh: float, perTile: bool) -> cwipc_pointcloud_wrapper:
    """Point cloud with statistical outliers removed."""
    return _wrap_filter_result('cwipc_remove_outliers',
                               cwipc_util_dll_load().cwipc_remove_outliers(pc.as_cwipc_p(), kNeighbors, stdDesvMultThresh, perTile))


def cwipc_tilefilter(pc: cwipc_pointcloud_wrapper, tile: int) -> cwipc_pointcloud_wrapper:
    """Only the points with the given tile number (0: all points)."""
    return _wrap_filter_result('cwipc_tilefilter', cwipc_util_dll_load().cwipc_tilefilter(pc.as_cwipc_p(), tile))


def cwipc_tilemap(pc: cwipc_pointcloud_wrapper, mapping: Union[List[int], Dict[int, int], bytes, bytearray]) -> cwipc_pointcloud_wrapper:
    """Every point's tile number replaced through a 256-entry table (list/bytes) or a dict of the entries to set."""
    if not isinstance(mapping, (bytes, bytearray, list)):
        m = [0] * 256
        for k in mapping:
            m[k] = mapping[k]
        mapping = m
    return _wrap_filter_result('cwipc_tilemap', cwipc_util_dll_load().cwipc_tilemap(pc.as_cwipc_p(), bytes(mapping)))


def cwipc_colormap(pc: cwipc_pointcloud_wrapper, clearBits: int, setBits: int) -> cwipc_pointcloud_wrapper:
    """Every point's packed colour word (tile<<24 | r<<16 | g<<8 | b) masked: word = (word & ~clearBits) | setBits."""
    return _wrap_filter_result('cwipc_colormap', cwipc_util_dll_load().cwipc_colormap(pc.as_cwipc_p(), clearBits, setBits))


def cwipc_crop(pc: cwipc_pointcloud_wrapper, bbox: Union[Tuple[float, float, float, float, float, float], List[float]]) -> cwipc_pointcloud_wrapper:
    """Points inside the half-open box minx, maxx, miny, maxy, minz, maxz."""
    return _wrap_filter_result('cwipc_crop', cwipc_util_dll_load().cwipc_crop(pc.as_cwipc_p(), (ctypes.c_float * 6)(*bbox)))


def cwipc_join(pc1: cwipc_pointcloud_wrapper, pc2: cwipc_pointcloud_wrapper) -> cwipc_pointcloud_wrapper:
    """Union of the two clouds: all points of pc1, then all points of pc2."""
    return _wrap_filter_result('cwipc_join', cwipc_util_dll_load().cwipc_join(pc1.as_cwipc_p(), pc2.as_cwipc_p()))


def cwipc_join_multi(pcs: Iterable[cwipc_pointcloud_wrapper]) -> cwipc_pointcloud_wrapper:
    """n-ary join.  The reference folds cwipc_join pairwise (util.py:1330-1332, copying O(n^2) bytes);
    this is one device pass with the same result (order, min timestamp, min cellsize)."""
    pcs = list(pcs)
    if not pcs:
        raise TypeError("cwipc_join_multi() of empty iterable")   # functools.reduce raises TypeError as well
    if len(pcs) == 1:
        return pcs[0]
    arr = (cwipc_pointcloud_p * len(pcs))(*[pc.as_cwipc_p() for pc in pcs])
    return _wrap_filter_result('cwipc_join_multi', cwipc_util_dll_load().cwipc_hip_join_multi(arr, len(pcs)))


# ---------------------------------------------------------------------------
# MI355X extensions
# ---------------------------------------------------------------------------
def cwipc_hip_device_count() -> int:
    return cwipc_util_dll_load().cwipc_hip_device_count()


def cwipc_hip_set_device(device: int) -> None:
    if cwipc_util_dll_load().cwipc_hip_set_device(device) != 0:
        raise CwipcError(f"cwipc_hip_set_device({device}) failed")


def cwipc_hip_pinned_points(npoints: int) -> cwipc_point_numpy_array_value_type:
    """A structured numpy array (cwipc_point dtype) of npoints records in page-locked memory (include/cwipc_util_amd/hip_ext.h,
    cwipc_hip_host_alloc): cwipc_from_numpy_array on it -- or on a slice of it -- is read by the GPU where it lies, without the
    copy into a staging buffer that ordinary host memory needs; copy_into() of a cloud writes it directly.  For buffers that are
    reused frame after frame (a capturer's output, a decoder's).  The memory is released with the array."""
    import weakref
    dll = cwipc_util_dll_load()
    nbytes = max(int(npoints), 1) * 16
    ptr = dll.cwipc_hip_host_alloc(nbytes)
    if not ptr:
        raise CwipcError("cwipc_hip_host_alloc failed (no GPU, or no page-locked memory left)")
    buf = (ctypes.c_byte * nbytes).from_address(ptr)
    arr = numpy.frombuffer(buf, dtype=cwipc_point_numpy_dtype, count=int(npoints))
    weakref.finalize(buf, dll.cwipc_hip_host_free, ptr)   # (arr.base keeps buf alive)
    return arr


class cwipc_hip_pin_array:
    """Page-lock the memory of an existing contiguous numpy array for as long as this object lives (cwipc_hip_host_register): the
    array can then be handed to cwipc_from_numpy_array / filled by copy_into() without a staging copy.  Use as a context manager or
    keep the object; registering costs about as much as copying the array a few times, so it pays for arrays that are reused."""

    def __init__(self, array: numpy.ndarray):
        if not array.flags['C_CONTIGUOUS']:
            raise ValueError("cwipc_hip_pin_array: the array must be contiguous")
        self._array = array
        self._ptr = array.ctypes.data
        if cwipc_util_dll_load().cwipc_hip_host_register(self._ptr, array.nbytes) != 0:
            self._ptr = None
            raise CwipcError("cwipc_hip_host_register failed")

    def release(self) -> None:
        if self._ptr:
            cwipc_util_dll_load().cwipc_hip_host_unregister(self._ptr)
            self._ptr = None

    def __enter__(self): return self._array
    def __exit__(self, *exc): self.release()
    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def cwipc_hip_upload(pc: cwipc_pointcloud_wrapper, drop_host_copy: bool = False) -> None:
    """Make the cloud device-resident now (filters do it lazily)."""
    dll = cwipc_util_dll_load()
    if dll.cwipc_hip_upload(pc.as_cwipc_p()) != 0:
        raise CwipcError("cwipc_hip_upload failed: " + dll.cwipc_hip_last_error().decode('utf8'))
    if drop_host_copy:
        dll.cwipc_hip_drop_host_copy(pc.as_cwipc_p())


def cwipc_hip_device_planes(pc: cwipc_pointcloud_wrapper) -> Tuple[int, int, int, int, int]:
    """Device addresses of the cloud's planes (x, y, z: float32[n]; rgbt: uint32[n] = r | g << 8 | b << 16 | tile << 24) and n.
    Uploads the cloud if it is not resident.  For zero-copy hand-over to other device code (and for tests that ask whether
    two clouds share their planes)."""
    x, y, z, w = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    n = ctypes.c_size_t(0)
    if cwipc_util_dll_load().cwipc_hip_device_planes(pc.as_cwipc_p(), ctypes.byref(x), ctypes.byref(y), ctypes.byref(z), ctypes.byref(w), ctypes.byref(n)) != 0:
        raise CwipcError("cwipc_hip_device_planes failed")
    return (x.value or 0, y.value or 0, z.value or 0, w.value or 0, n.value)


def cwipc_hip_colorize(pc: cwipc_pointcloud_wrapper, weight: float, lut: numpy.ndarray, valid: numpy.ndarray) -> cwipc_pointcloud_wrapper:
    """Device implementation of ColorizeFilter._mapcolor; lut (256,3) float64, valid (256,) bool."""
    lut = numpy.ascontiguousarray(lut, dtype=numpy.float64).reshape(256, 3)
    valid = numpy.ascontiguousarray(valid, dtype=numpy.uint8).reshape(256)
    rv = cwipc_util_dll_load().cwipc_hip_colorize(pc.as_cwipc_p(), float(weight), lut.ctypes.data, valid.ctypes.data)
    return _wrap_filter_result('cwipc_hip_colorize', rv)


def cwipc_hip_simulatecams(pc: cwipc_pointcloud_wrapper, camera_vectors: numpy.ndarray, centroid: numpy.ndarray) -> cwipc_pointcloud_wrapper:
    """Hard camera assignment of SimulatecamsFilter on the GPU: camera_vectors (ncam, 3) float64 with y = 0, centroid (3,) float32."""
    cams = numpy.ascontiguousarray(numpy.asarray(camera_vectors, dtype=numpy.float64)[:, [0, 2]])
    rv = cwipc_util_dll_load().cwipc_hip_simulatecams(pc.as_cwipc_p(), int(cams.shape[0]), float(numpy.float32(centroid[0])), float(numpy.float32(centroid[2])),
                                                      cams.ctypes.data)
    return _wrap_filter_result("cwipc_hip_simulatecams", rv)


CWIPC_HIP_COMM_ID_BYTES = 128
CWIPC_HIP_JOIN_LOOPBACK = 1


def cwipc_hip_comm_unique_id() -> bytes:
    """The id one rank makes and hands to the others (by any means) before they all call cwipc_hip_comm(id, rank, nranks)."""
    buf = ctypes.create_string_buffer(CWIPC_HIP_COMM_ID_BYTES)
    err = ctypes.c_char_p()
    if cwipc_util_dll_load().cwipc_hip_comm_unique_id(buf, ctypes.byref(err)) != 0:
        raise CwipcError("cwipc_hip_comm_unique_id: " + (err.value.decode('utf8') if err.value else "failed"))
    return buf.raw


class cwipc_hip_comm:
    """This rank's end of the multi-GPU join inside the library (RCCL; include/cwipc_util_amd/hip_ext.h): join(pc) once per
    frame on every rank, one C call, no torch on the way.  Creation is collective."""

    def __init__(self, unique_id: bytes, rank: int, nranks: int):
        if len(unique_id) != CWIPC_HIP_COMM_ID_BYTES:
            raise ValueError("cwipc_hip_comm: the id has %d bytes" % CWIPC_HIP_COMM_ID_BYTES)
        err = ctypes.c_char_p()
        self._dll = cwipc_util_dll_load()
        self._comm = self._dll.cwipc_hip_comm_create(unique_id, rank, nranks, ctypes.byref(err))
        if not self._comm:
            raise CwipcError("cwipc_hip_comm_create: " + (err.value.decode('utf8') if err.value else "failed"))
        self.rank, self.nranks = rank, nranks

    def join(self, pc: Optional[cwipc_pointcloud_wrapper], loopback: bool = False) -> cwipc_pointcloud_wrapper:
        """The fused cloud of this frame: every rank's points in rank order.  pc = None: no tile on this rank this frame."""
        if not self._comm:
            raise CwipcError("cwipc_hip_comm: used after free()")
        rv = self._dll.cwipc_hip_comm_join(self._comm, pc.as_cwipc_p() if pc is not None else None, CWIPC_HIP_JOIN_LOOPBACK if loopback else 0)
        return _wrap_filter_result('cwipc_hip_comm_join', rv)

    def submit(self, pc: Optional[cwipc_pointcloud_wrapper], loopback: bool = False) -> cwipc_pointcloud_wrapper:
        """join() for a stream of frames: returns at once; the fused cloud settles when it is first used (the exchange runs
        on a thread of the communicator, in the order of the calls).  pc may be freed right away."""
        if not self._comm:
            raise CwipcError("cwipc_hip_comm: used after free()")
        rv = self._dll.cwipc_hip_comm_submit(self._comm, pc.as_cwipc_p() if pc is not None else None, CWIPC_HIP_JOIN_LOOPBACK if loopback else 0)
        return _wrap_filter_result('cwipc_hip_comm_submit', rv)

    def free(self) -> None:
        if self._comm:
            self._dll.cwipc_hip_comm_free(self._comm)
            self._comm = None


def cwipc_tilefilter_masked(pc: cwipc_pointcloud_wrapper, mask: int) -> cwipc_pointcloud_wrapper:
    """Points whose tile number ANDed with mask is non-zero (reference python/cwipc/registration/util.py:98-112)."""
    return _wrap_filter_result('cwipc_tilefilter_masked', cwipc_util_dll_load().cwipc_hip_tilefilter_masked(pc.as_cwipc_p(), mask))


def cwipc_transform(pc: cwipc_pointcloud_wrapper, transform: Any) -> cwipc_pointcloud_wrapper:
    """Apply a 4x4 affine transformation (reference python/cwipc/registration/util.py:295-309: numpy
    `rot @ p + t` in float64, stored as float32).  Runs on the GPU, the cloud stays device-resident."""
    m = numpy.ascontiguousarray(numpy.asarray(transform, dtype=numpy.float64))
    if m.shape != (4, 4):
        raise ValueError("cwipc_transform: transform must be a 4x4 matrix")
    rv = cwipc_util_dll_load().cwipc_hip_transform(pc.as_cwipc_p(), m.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return _wrap_filter_result('cwipc_transform', rv)


def cwipc_offset_scale(pc: cwipc_pointcloud_wrapper, x: float, y: float, z: float, scale: float) -> cwipc_pointcloud_wrapper:
    """(p + (x, y, z)) * scale for every point, cellsize * scale (the loop of the reference's TransformFilter,
    python/cwipc/filters/transform.py:38-52, in the same float64 arithmetic)."""
    rv = cwipc_util_dll_load().cwipc_hip_offset_scale(pc.as_cwipc_p(), float(x), float(y), float(z), float(scale))
    return _wrap_filter_result('cwipc_offset_scale', rv)


def get_tiles_used(pc: cwipc_pointcloud_wrapper) -> List[int]:
    """Sorted list of the tile numbers that occur (reference python/cwipc/registration/util.py:285-293), without downloading the cloud."""
    used = (ctypes.c_ubyte * 256)()
    rc = cwipc_util_dll_load().cwipc_hip_tiles_used(pc.as_cwipc_p(), used)
    if rc < 0:
        raise CwipcError("get_tiles_used failed")
    return [t for t in range(256) if used[t]]


def cwipc_downsample_pertile(pc: cwipc_pointcloud_wrapper, cellsize: float) -> cwipc_pointcloud_wrapper:
    """Per-tile downsample, so points in different tiles are not combined (reference python/cwipc/registration/util.py:170-182):
    for every tile number that occurs, ascending, tilefilter -> downsample; the results joined in that order.  The reference
    folds pairwise joins; one n-ary join gives the same cloud (same order, ts = min, cellsize = min) in one pass."""
    tiles_used = get_tiles_used(pc)
    parts = [cwipc_downsample(cwipc_tilefilter(pc, tilenum), cellsize) for tilenum in tiles_used]
    assert parts
    return parts[0] if len(parts) == 1 else cwipc_join_multi(parts)


def cwipc_hip_knn_mean_dist(pc: cwipc_pointcloud_wrapper, kNeighbors: int, stddevMulThresh: float = 1.0) -> Tuple[numpy.ndarray, float]:
    """Intermediate result of remove_outliers: (d_i per point, threshold).  For parity tests."""
    n = pc.count()
    out = numpy.zeros(max(n, 1), dtype=numpy.float32)
    thr = ctypes.c_double(float('nan'))
    rc = cwipc_util_dll_load().cwipc_hip_knn_mean_dist(pc.as_cwipc_p(), kNeighbors, out.ctypes.data, out.size, ctypes.byref(thr), stddevMulThresh)
    if rc != 0:
        raise CwipcError("cwipc_hip_knn_mean_dist failed")
    return out[:n], float(thr.value)


def cwipc_hip_from_device_aos(dev_ptr: int, npoint: int, timestamp: int, cellsize: float) -> cwipc_pointcloud_wrapper:
    """New cloud from npoint cwipc_point records at a DEVICE address (e.g. torch tensor .data_ptr())."""
    rv = cwipc_util_dll_load().cwipc_hip_from_device_aos(dev_ptr, npoint, timestamp, cellsize)
    return _wrap_filter_result('cwipc_hip_from_device_aos', rv)


def cwipc_hip_from_device_slots(dev_ptr: int, slot_rows: int, header_rows: int, counts: List[int], timestamp: int, cellsize: float,
                                stream: Optional[int] = None) -> cwipc_pointcloud_wrapper:
    """New cloud from the receive buffer of an all-gather at a DEVICE address: len(counts) slots of slot_rows
    16-byte rows, the records of slot s in rows [header_rows, header_rows + counts[s]); slot order = point order.
    stream (a hipStream_t as an integer, e.g. torch.cuda.current_stream().cuda_stream): run as a step of that stream
    and return without waiting."""
    arr = (ctypes.c_uint32 * len(counts))(*counts)
    if stream is None:
        rv = cwipc_util_dll_load().cwipc_hip_from_device_slots(dev_ptr, len(counts), slot_rows, header_rows, arr, timestamp, cellsize)
    else:
        rv = cwipc_util_dll_load().cwipc_hip_from_device_slots_on_stream(dev_ptr, len(counts), slot_rows, header_rows, arr, timestamp, cellsize, stream)
    return _wrap_filter_result('cwipc_hip_from_device_slots', rv)


def cwipc_hip_copy_device_aos(pc: cwipc_pointcloud_wrapper, dev_ptr: int, size: int, stream: Optional[int] = None) -> int:
    """Interleave the cloud into a DEVICE buffer of `size` bytes; returns the number of points.  stream (a hipStream_t
    as an integer): run as a step of that stream and return without waiting."""
    if stream is None:
        n = cwipc_util_dll_load().cwipc_hip_copy_device_aos(pc.as_cwipc_p(), dev_ptr, size)
    else:
        n = cwipc_util_dll_load().cwipc_hip_copy_device_aos_on_stream(pc.as_cwipc_p(), dev_ptr, size, stream)
    if n < 0:
        raise CwipcError("cwipc_hip_copy_device_aos failed")
    return n


class cwipc_hip_profile:
    """Context manager collecting per-kernel device time (hipEvents on the library's stream).

        with cwipc_hip_profile() as prof:
            cwipc_downsample(pc, 0.01)
        prof.kernels  ->  {name: (total_ms, launches)}
    """

    def __enter__(self) -> 'cwipc_hip_profile':
        dll = cwipc_util_dll_load()
        dll.cwipc_hip_profile_reset()
        dll.cwipc_hip_profile_enable(1)
        self.kernels: Dict[str, Tuple[float, int]] = {}
        return self

    def __exit__(self, *exc) -> None:
        dll = cwipc_util_dll_load()
        dll.cwipc_hip_synchronize()
        dll.cwipc_hip_profile_enable(0)
        self.kernels = self.read()

    @staticmethod
    def read() -> Dict[str, Tuple[float, int]]:
        dll = cwipc_util_dll_load()
        out: Dict[str, Tuple[float, int]] = {}
        for i in range(dll.cwipc_hip_profile_count()):
            name, ms, cnt = ctypes.c_char_p(), ctypes.c_double(), ctypes.c_long()
            if dll.cwipc_hip_profile_get(i, ctypes.byref(name), ctypes.byref(ms), ctypes.byref(cnt)) == 0:
                out[name.value.decode('utf8')] = (ms.value, cnt.value)
        return out
