"""vrdone_amd: MI355X-native (gfx950) implementation of VrdONE's relation-encoding hot path.

``vrdone_amd.models`` mirrors the reference's ``models`` package surface (same class names,
constructor signatures and parameter trees) with every forward running hand-written HIP
kernels through ``libvrdone_hip.so``.  There is no CPU execution path.
"""
__version__ = "0.1.0"
