from .prior import Prior, NormalPrior, UniformPrior
