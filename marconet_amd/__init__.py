"""marconet_amd — MI355X-native (gfx950) implementation of MARCONet's test_sr.py / test_w.py inference forward.

Python host (same nn.Module names, state_dict keys and forward() signatures as the reference's
models/networks.py and models/textvit_arch.py) over hand-written HIP kernels behind a C-ABI
(include/marconet_hip.h, marconet_amd/lib/libmarconet_hip.so).  No CPU / eager fallback exists.
"""
__version__ = "0.1.0"
