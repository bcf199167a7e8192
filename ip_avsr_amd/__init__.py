"""ip_avsr_amd -- MI355X-native AdeNet / DeltaNet training path.

Importing the package is cheap and CPU-safe (host-side utilities only).  The HIP compute
library (``ip_avsr_amd/csrc/libadenet_hip.so``) is loaded lazily by ``ip_avsr_amd._lib`` the first
time a model is created, and that load fails loudly if the library is missing: there is no CPU
fallback for the compute path.
"""
__version__ = "0.1.0"
