"""Callers either side of the filter path: the tile synchroniser."""
from .source_synchronizer import cwipc_source_synchronizer, SyncCore

__all__ = ["cwipc_source_synchronizer", "SyncCore"]
