"""MI355X-native kernel matrix-vector product backend (CDNA4 / gfx950).

Host side of ``libkmvp.so`` (hand-written HIP kernels behind the C ABI of
``include/kmvp.h``) and the mirror of the kernel-matrix-benchmarks plugin
interface for this path.  Importing this package never touches the GPU.
"""
__version__ = "0.1.0"
