"""cfdbench_amd: MI355X-native (gfx950) hot path for CFDBench's autoregressive neural operators.

Importing this package does not load the HIP extension; the first compute call does (``cfdbench_amd._lib.api()``)
and raises if ``cfdbench_amd/_C/libcfdbench_amd.so`` has not been built -- there is no CPU / ATen fallback.
"""
__version__ = "0.1.0"
