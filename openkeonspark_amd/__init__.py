"""openkeonspark_amd: MI355X-native engine for the OpenKEonSpark training hot path.

`Config` + the `TransE/TransH/TransR/TransD` model classes keep the reference's Python API
(/root/reference/Config.py, Model.py); the work runs in hand-written HIP kernels behind the C ABI of
`csrc/libkge_mi355.so` (include/kge_mi355.h).  Importing the package does not load the shared
library (so `__graft_entry__.build()` can import it before the first build); constructing a
`Config` does, and fails loudly if it is missing.
"""
from ._lib import KgeError  # noqa: F401
from .Config import Config  # noqa: F401
from .TransE import TransE  # noqa: F401
from .TransH import TransH  # noqa: F401
from .TransR import TransR  # noqa: F401
from .TransD import TransD  # noqa: F401
