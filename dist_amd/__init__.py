"""dist_amd — MI355X-native DiST training hot path (frozen CLIP ViT + DiST branch).

Importing the package is cheap and GPU-free; the HIP library is loaded lazily by
dist_amd.lib.load() and every compute entry point fails loudly without it.
"""
__version__ = "0.1.0"
