"""`import r3d` -> the product package (whose directory name, 3d_reconstruction_project_amd, is not an identifier)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module("3d_reconstruction_project_amd")
