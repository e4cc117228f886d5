"""diskrag_amd -- MI355X-native Vamana beam-search engine behind the search seam of Jolara-ai/diskrag.

Host glue in Python (ctypes); everything on the query path runs in libdiskrag_hip.so (hand-written HIP for
gfx950). Importing the package does not load the library; the first engine call does, and raises if it is missing.
"""
from . import _ffi  # noqa: F401
from . import persist  # noqa: F401
from ._ffi import HipIndex, DiskragHipError, device_count, load_library  # noqa: F401
from .search_engine import SearchEngineCorrect, SearchEngine, export_codebook  # noqa: F401
from .batching import RequestBatcher  # noqa: F401

__all__ = ["HipIndex", "DiskragHipError", "device_count", "load_library", "SearchEngineCorrect", "SearchEngine",
           "export_codebook", "persist", "RequestBatcher"]
