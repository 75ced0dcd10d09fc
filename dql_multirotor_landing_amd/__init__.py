"""MI355X-native vectorised UAV-landing environment + tabular Double-Q trainer (hot path of
valerio98-lab/DQL_multirotor_landing).  Host side: thin ctypes binding over the HIP library
`csrc/libdql_hip.so` (C ABI in include/dql.h)."""
from .config import DqlConfig, simulation_config, training_config  # noqa: F401
