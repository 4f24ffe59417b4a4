"""bayesianinference_amd -- MI355X-native Gaussian-process likelihood / prediction path.

Drop-in for the GP hot path of ssmit1986/BayesianInference: hand-written HIP
kernels for gfx950 behind a plain C ABI (include/gphip.h), with a Python host-side mirror of the
reference's `defineGaussianProcess` / `predictFromGaussianProcess` / `inferenceObject` interface.
"""
__version__ = "0.5.0"
