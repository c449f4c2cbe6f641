from .volatility_likelihood import VolatilityGaussianLikelihood      # voltron/likelihoods/__init__.py
