"""Importable alias of the package directory ``sif-xco2-cokriging_amd/`` (a hyphen
cannot appear in a Python module name).  All code lives in that directory; this
file only points the import system at it:

    from sif_xco2_cokriging_amd import joint_prediction as prediction
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "sif-xco2-cokriging_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
