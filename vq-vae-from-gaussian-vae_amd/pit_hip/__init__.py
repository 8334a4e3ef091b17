"""pit_hip -- MI355X-native mirror of the reference's `pit` package for the
encode -> quantize -> decode hot path.  The heavy lifting is in
libgqhip.so (csrc/), bound through `pit_hip._lib`."""

__all__ = ["_lib"]
