"""Drop-in alias: `import pycbinfer` resolves to the MI355X implementation in `cbinfer_amd`.

The reference's applications import `pycbinfer`, reach the classes as `pycbinfer.CBConv2d` and
`pycbinfer.conv2d.CBConv2d` (sceneLabeling/modelConverter.py:33, poseDetection/evalTools.py:91,101),
and its pickled models name `pycbinfer.conv2d.CBConv2d` / `CBPoolMax2d`; all of these resolve here.
"""
import sys

import cbinfer_amd
from cbinfer_amd import *            # noqa: F401,F403
from cbinfer_amd import conv2d, conv2d_cg, conv2d_fg

sys.modules[__name__ + '.conv2d'] = conv2d
sys.modules[__name__ + '.conv2d_cg'] = conv2d_cg
sys.modules[__name__ + '.conv2d_fg'] = conv2d_fg

__all__ = cbinfer_amd.__all__
