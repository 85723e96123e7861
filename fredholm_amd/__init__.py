"""fredholm_amd -- MI355X (gfx950) native replacement for the render loop of yumcyaWiz/fredholm.

The product is `libfredholm_hip.so` (hand-written HIP kernels behind the C ABI of
include/fredholm_hip.h).  This package is the thin Python host side: a ctypes binding
(`fredholm_amd.native`), a mirror of the reference's `fredholm::Renderer` interface
(`fredholm_amd.renderer`) and the synthetic scene generators used by tests and bench
(`fredholm_amd.scenes`).  There is no CPU fallback: without the built library or without a GPU
every compute call raises.
"""
from .native import FredholmError, lib, load_library  # noqa: F401
from .renderer import Camera, RenderLayer, Renderer, PostProcessParams  # noqa: F401
from . import scenes  # noqa: F401
