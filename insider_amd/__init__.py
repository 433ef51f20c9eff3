"""insider_amd — MI355X-native INSIDER factorisation core (HIP kernels behind a C-ABI)."""
__version__ = "0.1.0"
