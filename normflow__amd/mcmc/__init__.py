from .mcmc import MCMCSampler, MCMCHistory, Metropolis
