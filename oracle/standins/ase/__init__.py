"""TEST-ONLY stand-in for ase: only ase.data.atomic_masses is read (schnet.py:47)."""
from . import data  # noqa: F401
