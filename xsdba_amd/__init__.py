"""xsdba_amd — MI355X-native implementation of xsdba's empirical-quantile-mapping hot path.

Compute runs in hand-written HIP kernels (``xsdba_amd/csrc``) behind the C ABI of ``include/xsdba_hip.h``;
this package is the host-side mirror of the reference's interface for that path
(``TrainAdjust`` / ``map_groups`` / ``Grouper``).  There is no CPU fallback: without ``libxsdba_hip.so`` and an
MI355X the compute calls raise.
"""
__version__ = "0.1.0"
