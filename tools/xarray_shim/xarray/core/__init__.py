"""container-only stand-in, see ../__init__.py"""
from . import dataarray  # noqa: F401
