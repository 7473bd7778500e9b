"""diffbir_amd — MI355X-native DiffBIR inference hot path (HIP kernels behind a C ABI)."""
__version__ = "0.1.0"
