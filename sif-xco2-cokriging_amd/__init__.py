"""MI355X-native bivariate Matern cokriging: drop-in for the hot path of
91Mrwu/sif-xco2-cokriging (joint_prediction / point_prediction / the covariance part
of model / distance + variogram part of fields).

Python here is host plumbing only; every number comes from hand-written HIP kernels
in ``csrc/`` through the C ABI declared in ``include/cokrige.h``
(``libcokrige_hip.so``).  There is no CPU fallback: without the library or without a
GPU the numeric entry points raise.
"""
__all__ = ["joint_prediction", "point_prediction", "model", "fields", "sim", "native"]
__version__ = "0.1.0"
