"""container-only stand-in, see ../../__init__.py"""
from .. import DataArray  # noqa: F401  (same class object)
