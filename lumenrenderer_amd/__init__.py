"""lumenrenderer_amd — MI355X-native wavefront path tracer behind the LumenRenderer plugin API.

The product is the C-ABI shared library ``liblumen_mi.so`` (hand-written HIP kernels for gfx950 + C++ host code,
sources in ``csrc/``, interface in ``include/lumen_mi.h``).  This package is the thin Python host layer used by the
tests, ``bench.py`` and the multi-GPU tile driver: ctypes bindings (``capi``), a mirror of the reference's
``LumenRenderer`` call surface (``renderer``), scene descriptions / procedural benchmark scenes (``scenes``), a glTF
reader that follows the reference's ``.ollad`` ingest rules (``gltf``) and tile sharding over ``torch.distributed``
(``tiles``).  There is no CPU or PyTorch fallback: if the library is missing or no GPU is present, rendering calls
raise.
"""
from .capi import LumenMIError, library_path, load_library  # noqa: F401
from .renderer import LumenRendererMI  # noqa: F401
from .scenes import SceneDescription  # noqa: F401

__all__ = ["LumenRendererMI", "SceneDescription", "LumenMIError", "load_library", "library_path"]
