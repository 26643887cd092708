"""colorize filter: recolour points by tile number or camera mask.

Reference python/cwipc/filters/colorize.py: the colour maps (:15-55) and the
constructor's argument handling (:68-83) are restated here; the per-point blend
(:100-119, a Python loop over ctypes structs there) runs in a HIP kernel
(cwipc_hip_colorize) that evaluates the same IEEE-double expression per channel:

    new = int((colour * weight + (old / 255.0) * (1 - weight)) * 255)
"""
from typing import Any, Dict, List, Optional, Tuple

import numpy

from .abstract import _TimedFilter
from ..util import cwipc_hip_colorize, cwipc_pointcloud_wrapper

ColorTuple = Tuple[float, float, float]


class ColorMap:
    """256-entry tile -> colour table; tiles without an entry keep their colour."""

    def __init__(self, initializer: Optional[Dict[int, ColorTuple]] = None):
        self._map: List[Optional[ColorTuple]] = [None] * 256
        self._tables: Optional[Tuple[numpy.ndarray, numpy.ndarray]] = None   # what tables() made of the map as it stands
        if initializer:
            for k, v in initializer.items():
                self._map[k] = v

    def add_mapping(self, tilenum: int, color: ColorTuple) -> None:
        self._map[tilenum] = color
        self._tables = None

    def map(self, tilenum: int) -> Optional[ColorTuple]:
        return self._map[tilenum]

    def tables(self) -> Tuple[numpy.ndarray, numpy.ndarray]:
        """(lut (256,3) float64, valid (256,) uint8) -- the form the C entry point takes.  Made once per state of the map: a filter
        colours every tile of every frame with the same map, and the loop below was a third of a call on a camera tile."""
        if self._tables is not None:
            return self._tables
        lut = numpy.zeros((256, 3), dtype=numpy.float64)
        valid = numpy.zeros(256, dtype=numpy.uint8)
        for t, c in enumerate(self._map):
            if c is not None:
                lut[t] = [float(c[0]), float(c[1]), float(c[2])]
                valid[t] = 1
        lut.setflags(write=False)
        valid.setflags(write=False)
        self._tables = (lut, valid)
        return self._tables


# reference colorize.py:21-29 -- one colour per single-camera tile number
_colorMapTiles = ColorMap({
    1: (1, 0, 0), 2: (0, 1, 0), 4: (0, 0, 1), 8: (0.5, 0.5, 0),
    16: (0, 0.5, 0.5), 32: (0.5, 0, 0.5), 64: (0.2, 0.2, 0.2), 128: (0.7, 0.7, 0.7),
})

# reference colorize.py:31-50 -- colour by number of contributing cameras; note range(255): tile 255 has no entry
_colorForBitCount = [(0.2, 0.2, 0.2), (1, 1, 1), (1, 0, 0), (0, 1, 0), (0, 0, 1), (0.5, 0.5, 0), (0, 0.5, 0.5), (0.5, 0, 0.5), (0, 0, 0)]
_colorMapContributingCameras = ColorMap()
for _i in range(255):
    _colorMapContributingCameras.add_mapping(_i, _colorForBitCount[bin(_i).count('1')])

_namedColorMaps = dict(camera=_colorMapTiles, contributions=_colorMapContributingCameras)


class ColorizeFilter(_TimedFilter):
    """
    colorize - Change the color of points in a pointcloud, based on the tile number or mask.
        Arguments:
            weight: 1.0 means completely replace original color, 0.0 changes nothing
            colormap: a 3-float-tuple for a uniform color, otherwise a colorize.ColorMap or the name of one:
                      camera: Each tile number gets a different color
                      contributions: the color depends on the number of bits set in the tilenumber
    """
    filtername = "colorize"

    def __init__(self, weight: float, colormap: Any):
        super().__init__()
        if isinstance(colormap, str) and colormap in _namedColorMaps:
            self.colorMap = _namedColorMaps[colormap]
        elif type(colormap) == type(()):
            self.colorMap = ColorMap()
            for i in range(256):
                self.colorMap.add_mapping(i, colormap)
        elif isinstance(colormap, ColorMap):
            self.colorMap = colormap
        else:
            self.colorMap = ColorMap(colormap)
        self.weight = weight

    def filter(self, pc: cwipc_pointcloud_wrapper) -> cwipc_pointcloud_wrapper:
        return self._run(pc, self._mapcolor)

    def _mapcolor(self, pc: cwipc_pointcloud_wrapper) -> cwipc_pointcloud_wrapper:
        lut, valid = self.colorMap.tables()
        return cwipc_hip_colorize(pc, self.weight, lut, valid)

    def statistics(self) -> None:
        if self.times:
            self.print1stat('duration', self.times)
        if self.original_pointcounts:
            self.print1stat('original_pointcount', self.original_pointcounts, True)


CustomFilter = ColorizeFilter
