"""Module name the reference's train prototxt refers to (README.md:57-76: ``module: "data_argumentation_layer"``,
``layer: "DataArgumentationLayer"``); with ``fcn_object_detector_amd/python`` on PYTHONPATH the same prototxt resolves here."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from fcn_object_detector_amd.data_layer import DataArgumentationLayer  # noqa: E402,F401
