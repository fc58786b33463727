"""davo_amd — MI355X-native frame-to-frame pose inference path of DAVO."""
from .launch import bind_rank_cpus as _bind_rank_cpus      # imports os and the standard library only
_bind_rank_cpus()        # a rank started by davo_amd.launch binds itself to its CPU slice before numpy or HIP start a thread
from .version import parse_version, FLAGSHIP_VERSION, VariantConfig, UnsupportedVariantError  # noqa: F401
from .davo import DAVO, Engine, DavoError, DavoRangeError, conv2d_same, pinned_empty, pin_array, unpin_array  # noqa: F401
