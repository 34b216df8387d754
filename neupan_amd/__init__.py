"""neupan_amd -- MI355X-native PAN inner solver (drop-in for neupan.blocks.PAN).

    from neupan_amd import neupan, PAN, Robot, FleetPlanner, NominalBatch, scan_to_point_batch, DuneTrain
"""
from .robot import Robot  # noqa: F401
from .scenes import CONFIGS, SceneConfig, make_batch, make_scene  # noqa: F401


_LAZY = {"PAN": ("pan", "PAN"), "forward_interleaved": ("pan", "forward_interleaved"),
         "FleetPlanner": ("fleet", "FleetPlanner"), "NominalBatch": ("frontend", "NominalBatch"),
         "scan_to_point_batch": ("frontend", "scan_to_point_batch"),
         "scan_to_point_velocity_batch": ("frontend", "scan_to_point_velocity_batch"),
         "DuneTrain": ("dune_train", "DuneTrain"), "neupan": ("planner", "neupan")}      # (dune_labels.dune_labels: import it from its module)


def __getattr__(name):
    # these pull in torch + the HIP library; keep `import neupan_amd.scenes` light for CPU tools
    if name in _LAZY:
        import importlib
        mod, attr = _LAZY[name]
        return getattr(importlib.import_module("." + mod, __name__), attr)
    raise AttributeError(name)
