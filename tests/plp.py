"""Import helper: the package directory name contains a hyphen."""
import importlib
plp = importlib.import_module("structure-plp-slam_amd")
synth = importlib.import_module("structure-plp-slam_amd.synth")
