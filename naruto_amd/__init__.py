"""naruto_amd -- MI355X-native (gfx950) implementation of NARUTO's neural-implicit mapping / uncertainty
hot path behind the reference's own operator surface.  See DESIGN.md / INTEGRATION.md.

Importing the package does not need a GPU; the first operator call loads libnaruto_hip.so (built
in-tree by ``__graft_entry__.build()``) and fails loudly if it is missing.
"""

__version__ = "0.1.0"

from . import config, synthetic  # noqa: F401


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `import naruto_amd` stays cheap
    if name in ("ops", "field", "parallel", "trainer", "_lib", "graphed", "ba_loop", "keyframe_store", "active_ray_sampler", "planner_aggregation", "mesh"):
        import importlib
        return importlib.import_module("." + name, __name__)
    if name == "NarutoFieldHIP":
        from .field import NarutoFieldHIP
        return NarutoFieldHIP
    if name in ("MappingTrainer", "FusedAdam"):
        from . import trainer
        return getattr(trainer, name)
    if name == "FusedBA":
        from .ba_loop import FusedBA
        return FusedBA
    raise AttributeError(name)
