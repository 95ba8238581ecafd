"""MI355X-native implementation of Kaldi's acoustic-scoring + lattice-decoding hot
path.  Import with importlib.import_module("old-kaldi-git_amd").

Nothing here computes on the CPU: `api` calls libkaldi_hip.so through its C-ABI
(include/kaldi_hip.h) and raises if the library is missing or no gfx950 device
is usable."""
from . import capi  # noqa: F401


def load_library():
    return capi.load()
