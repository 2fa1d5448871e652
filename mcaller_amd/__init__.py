"""mcaller_amd -- MI355X-native m6A caller: the hot path of al-mcintyre/mCaller as HIP kernels.

Host side (Python, like the reference): `extract_contexts.extract_features` (drop-in signature),
`read_qual.extract_read_quality`, `mCaller.main` (CLI).  Device side: libmcaller_hip.so through ctypes
(`_lib`, `device`).  No PyTorch, no CPU fallback.
"""
__version__ = '0.1'
