"""strikeforce_amd — MI355X-native batched StrikeForce arena simulator (host-side Python plumbing).

The product path is the HIP library behind include/strikeforce.h; see ``env.ArenaBatch``.
"""
from . import abi, config  # noqa: F401
