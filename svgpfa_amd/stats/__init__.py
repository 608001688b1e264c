"""Host-side mirror of the reference's svGPFA.stats API surface over the HIP engine.

Same class / method names, argument meaning and error behaviour as
/root/reference/src/svGPFA/stats/{kernels, kernelsMatricesStore, svLowerBound, svGPFAModelFactory, svEM}.py,
so that callers written against the reference (its EM loop, its plotting cells) run unchanged.
All numerical work happens in libsvgpfa_hip.so; there is no CPU fallback."""
