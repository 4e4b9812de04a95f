"""Importable alias of the ``modelardb-rs_amd`` package (a hyphen cannot appear in ``import``)."""

import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.modules[__name__] = importlib.import_module("modelardb-rs_amd")
