"""Which object executes the device calls of the plugin classes.

The default backend is the HIP library (``_lib.default_context()``): it is
created on first use and raises if ``libdynhip.so`` or a GPU is missing --
there is no CPU fallback in the product.  ``set_backend`` exists so that the
test-suite can exercise the host-side plumbing (pickling, dynesty integration)
on machines without a GPU by injecting an object with the same methods
(tests/oracle_backend.py); nothing in the package does that by itself.
"""
_active = None
_device = 0


def set_device(device):
    """Select the GPU ordinal used by the default backend (before first use)."""
    global _device, _active
    _device = int(device)
    _active = None


def set_backend(obj):
    global _active
    _active = obj


def get_backend():
    global _active
    if _active is None:
        from . import _lib
        _active = _lib.default_context(_device)
    return _active
