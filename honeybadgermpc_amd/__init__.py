"""
honeybadgermpc_amd -- MI355X-native batch share reconstruction for HoneyBadgerMPC.

A drop-in for one hot path of initc3/HoneyBadgerMPC: GF(p) polynomial
evaluation / interpolation and Reed-Solomon encode / decode
(reference: honeybadgermpc/ntl, polynomial.py, reed_solomon.py,
reed_solomon_wb.py, batch_reconstruction.py).  Host code is Python with the
reference's names; the arithmetic runs in hand-written HIP kernels for gfx950
reached through the C ABI in include/hbmpc_hip.h.  There is no CPU fallback.
"""
__version__ = "0.1.0"
