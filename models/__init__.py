"""Drop-in ``models`` package: the import surface of the reference's models/ directory, served by the MI355X-native
implementation in ``marconet_amd``.

The reference's scripts start with ``from models import networks, ocr`` (test_sr.py:6) / ``from models import networks``
(test_w.py:6) and build ``networks.TSPGAN()``, ``networks.TSPSRNet()``, ``networks.TextContextEncoderV2()`` before
``load_state_dict(torch.load(...)['params'], strict=True)`` (test_sr.py:42-52).  With this directory on ``sys.path`` in place
of the reference's own models/ those lines run unmodified and every forward goes through libmarconet_hip.so.

Importing the package also registers the ``basicsr.ops.fused_act`` provider (models/networks.py:10 of the reference imports
it; upstream basicsr ships it as a CUDA-only extension), so third-party code that imports it keeps working.
"""
from marconet_amd import fused_act as _fused_act

_fused_act.install()
