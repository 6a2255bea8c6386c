from .prior import Prior, NormalPrior, UniformPrior, PriorList, BlockUpdater
