"""
snpmatch_amd -- MI355X-native scoring engine for SNPmatch's Genotyper / CrossIdentifier hot path.

Host-side mirror of the reference's Python interface (snpmatch_amd.core.snpmatch, .csmatch, ...)
over a C-ABI shared library of hand-written HIP kernels (include/snpmatch_hip.h).
"""
__version__ = "0.1.0"
