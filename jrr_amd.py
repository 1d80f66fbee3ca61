"""Importable alias for the `joint-regressor-refinement_amd` package (hyphenated directory name)."""
import importlib
import sys

sys.modules[__name__] = importlib.import_module('joint-regressor-refinement_amd')
