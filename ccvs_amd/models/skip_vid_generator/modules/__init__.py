"""Host-side mirrors of the reference's L2 ops (modules/__init__.py:13-14), each a thin
call into libccvs_hip.so."""
from .upfirdn2d import upfirdn2d
from .correlation import FunctionCorrelation
from .quantize import VectorQuantizer
