"""openess_amd -- MI355X-native (gfx950) implementation of the OpenESS data-parallel hot path.

Hand-written HIP kernels behind a C-ABI (include/oess.h, openess_amd/liboess.so) with a Python
host layer that mirrors the reference's module / function names."""
__version__ = "0.1.0"
