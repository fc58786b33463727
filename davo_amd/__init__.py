"""davo_amd — MI355X-native frame-to-frame pose inference path of DAVO."""
from .version import parse_version, FLAGSHIP_VERSION, VariantConfig, UnsupportedVariantError  # noqa: F401
from .davo import DAVO, Engine, DavoError, DavoRangeError, conv2d_same, pinned_empty, pin_array, unpin_array  # noqa: F401
