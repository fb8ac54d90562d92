"""leaffliction_amd — MI355X-native hot path of leaffliction.

Layout mirrors the reference's `srcs/` packages for the parts on the hot path
(preprocessing, dataio, model, train, predict, cli); `csrc/` holds the HIP kernels and
the C ABI (`include/leafhip.h`), `_lib.py` the ctypes binding, `ops.py` / `nn.py` the
tensor-level launchers.
"""
__version__ = "0.1.0"
