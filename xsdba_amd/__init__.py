"""xsdba_amd — MI355X-native implementation of xsdba's empirical-quantile-mapping hot path.

Compute runs in hand-written HIP kernels (``xsdba_amd/csrc``) behind the C ABI of ``include/xsdba_hip.h``;
this package is the host-side mirror of the reference's interface for that path
(``TrainAdjust`` / ``map_groups`` / ``Grouper``), so that

    from xsdba_amd import EmpiricalQuantileMapping, Grouper
    EQM = EmpiricalQuantileMapping.train(ref, hist, group=Grouper("time.dayofyear", window=31), nquantiles=20)
    scen = EQM.adjust(sim)

reads like the reference.  There is no CPU fallback: without ``libxsdba_hip.so`` and an MI355X the compute
calls raise.
"""
__version__ = "0.1.0"

from . import _adjustment, adjustment, base, calendar, detrending, nbutils, options, processing, utils  # noqa: E402,F401
from .adjustment import (  # noqa: E402,F401
    BaseAdjustment,
    DetrendedQuantileMapping,
    EmpiricalQuantileMapping,
    LOCI,
    QuantileDeltaMapping,
    Scaling,
    TrainAdjust,
)
from .base import Grouper, map_blocks, map_groups, parse_group  # noqa: E402,F401
from .calendar import TimeIndex, date_range  # noqa: E402,F401
from .detrending import MeanDetrend, NoDetrend, PolyDetrend, RollingMeanDetrend  # noqa: E402,F401
from .labeled import DataArray, Dataset, as_dataarray  # noqa: E402,F401
from .options import set_options  # noqa: E402,F401
from .io import open_dataset, to_netcdf  # noqa: E402,F401
