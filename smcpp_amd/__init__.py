

# PyTorch (device memory, streams, torch.distributed: plumbing) bundles its own HIP runtime whose SONAME equals the one of
# /opt/rocm/lib that libsmcpp_engine.so is linked against.  Whichever copy is loaded first serves the whole process; a process
# that loads the engine first and torch's runtime second ends up with two, and the second cannot open the GPU ("No HIP GPUs
# are available").  Importing torch here - before any engine library can be loaded - makes its runtime the only one.
try:
    import torch as _torch  # noqa: F401
except Exception:  # noqa: BLE001  (the engine itself does not need torch)
    _torch = None
