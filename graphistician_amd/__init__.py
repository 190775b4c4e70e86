"""graphistician_amd — MI355X-native Gibbs engine behind graphistician's
networks.py surface (resample / log_likelihood).  See DESIGN.md.

Mirrors `graphistician/__init__.py:2`: the four public modules.
"""
from . import abstractions, adjacency, weights, networks, eigenmodel, ais  # noqa: F401
