"""openkeonspark_amd: MI355X-native engine for the OpenKEonSpark training hot path.

`Config` + the `TransE/TransH/TransR/TransD` model classes keep the reference's Python API
(/root/reference/Config.py, Model.py); the work runs in hand-written HIP kernels behind the C ABI of
`csrc/libkge_mi355.so` (include/kge_mi355.h).
"""
from ._lib import KgeError  # noqa: F401


def __getattr__(name):
    # lazy: importing the package must not require the built library (build() imports it first)
    if name == "Config":
        from .Config import Config
        return Config
    if name in ("TransE", "TransH", "TransR", "TransD"):
        import importlib
        return getattr(importlib.import_module("." + name, __name__), name)
    raise AttributeError(name)
