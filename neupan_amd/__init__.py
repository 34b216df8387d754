"""neupan_amd -- MI355X-native PAN inner solver (drop-in for neupan.blocks.PAN).

    from neupan_amd import PAN, Robot
"""
from .robot import Robot  # noqa: F401
from .scenes import CONFIGS, SceneConfig, make_batch, make_scene  # noqa: F401


def __getattr__(name):
    # PAN pulls in torch + the HIP library; keep `import neupan_amd.scenes` light for CPU tools
    if name == "PAN":
        from .pan import PAN
        return PAN
    raise AttributeError(name)
