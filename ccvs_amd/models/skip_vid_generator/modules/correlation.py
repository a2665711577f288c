"""`FunctionCorrelation(first, second, stride)` with the reference's signature
(modules/correlation.py:405-406), forward only, on the HIP cost-volume kernel."""
from ccvs_amd import ops


def FunctionCorrelation(tenFirst, tenSecond, stride):
    assert tenFirst.is_contiguous() and tenSecond.is_contiguous()  # modules/correlation.py:291-292
    return ops.correlation7x7(tenFirst, tenSecond, stride)
