"""TEST-ONLY: the reference's tests parametrise over its two compilers; both names
resolve to the device plugin here."""
from triflow import device_compiler as numpy_compiler      # noqa: F401
from triflow import device_compiler as theano_compiler     # noqa: F401
