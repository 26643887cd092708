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
    @abstractmethod
    def free(self) -> None: ...

    @abstractmethod
    def eof(self) -> bool: ...

    @abstractmethod
    def available(self, wait: bool) -> bool: ...

    @abstractmethod
    def get(self) -> Optional[cwipc_pointcloud_abstract]: ...

    @abstractmethod
    def statistics(self) -> None: ...


class cwipc_activesource_abstract(cwipc_source_abstract):
    @abstractmethod
    def reload_config(self, config: Union[str, bytes, None]) -> Any: ...

    @abstractmethod
    def get_config(self) -> bytes: ...

    @abstractmethod
    def start(self) -> bool: ...

    @abstractmethod
    def stop(self) -> None: ...

    @abstractmethod
    def seek(self, timestamp: int) -> bool: ...

    @abstractmethod
    def request_metadata(self, name: str) -> None: ...

    @abstractmethod
    def is_metadata_requested(self, name: str) -> bool: ...

    @abstractmethod
    def auxiliary_operation(self, op: str, inbuf: bytes, outbuf: bytearray) -> bool: ...

    @abstractmethod
    def maxtile(self) -> int: ...

    @abstractmethod
    def get_tileinfo_dict(self, tilenum: int) -> cwipc_tileinfo_dict: ...
