"""Abstract interfaces of point clouds and sources (same method sets as reference python/cwipc/abstract.py)."""
from abc import ABC, abstractmethod
from typing import Any, Dict, Optional, Union

cwipc_tileinfo_dict = Dict[str, Any]


class cwipc_pointcloud_abstract(ABC):
    @abstractmethod
    def free(self) -> None:
        """Release the native point cloud."""

    @abstractmethod
    def timestamp(self) -> int:
        """Capture timestamp of the cloud."""

    @abstractmethod
    def cellsize(self) -> float:
        """Size of the cells the points represent (0 if unknown)."""

    @abstractmethod
    def count(self) -> int:
        """Number of points."""


class cwipc_source_abstract(ABC):
    """Something that hands out point clouds one at a time (a generator, a decoder, a synchroniser)."""

    @abstractmethod
    def free(self) -> None:
        """Give the native object back; the wrapper is unusable afterwards."""

    @abstractmethod
    def eof(self) -> bool:
        """True once no further cloud will ever come."""

    @abstractmethod
    def available(self, wait: bool) -> bool:
        """Is a cloud ready to be taken?  With wait=True the call may block until one is (or until end of stream)."""

    @abstractmethod
    def get(self) -> Optional[cwipc_pointcloud_abstract]:
        """The next cloud, or None when there is none to be had."""

    @abstractmethod
    def statistics(self) -> None:
        """Print whatever the source counted while it ran."""


class cwipc_activesource_abstract(cwipc_source_abstract):
    """A source with a life cycle of its own (cameras, the synthetic generator): it is started and stopped, can be
    configured, knows its tiles, and may attach metadata to the clouds it produces."""

    @abstractmethod
    def start(self) -> bool:
        """Begin producing; False if that is not possible."""

    @abstractmethod
    def stop(self) -> None:
        """Stop producing."""

    @abstractmethod
    def reload_config(self, config: Union[str, bytes, None]) -> Any:
        """Apply a new configuration (JSON text, a file name, or None for the default)."""

    @abstractmethod
    def get_config(self) -> bytes:
        """The configuration in force, as JSON."""

    @abstractmethod
    def seek(self, timestamp: int) -> bool:
        """Move a recorded stream to the given timestamp; False where that means nothing."""

    @abstractmethod
    def maxtile(self) -> int:
        """Number of tile descriptions get_tileinfo_dict() can answer for."""

    @abstractmethod
    def get_tileinfo_dict(self, tilenum: int) -> cwipc_tileinfo_dict:
        """Description of one tile: its normal, its camera, how many cameras contribute, the camera mask."""

    @abstractmethod
    def request_metadata(self, name: str) -> None:
        """Ask for the named per-frame metadata to be attached to every cloud from now on."""

    @abstractmethod
    def is_metadata_requested(self, name: str) -> bool:
        """Has request_metadata(name) been called?"""

    @abstractmethod
    def auxiliary_operation(self, op: str, inbuf: bytes, outbuf: bytearray) -> bool:
        """Source-specific side channel: operation name, input bytes, room for the answer; False if unknown."""
