"""Drop-in for the reference's kernel/compression.py: same two host functions (convert_key_batched,
convert_value_batched), computed by mustafar_amd's HIP kernels."""
from mustafar_amd.compression import convert_key_batched, convert_value_batched, prune_magnitude  # noqa: F401
