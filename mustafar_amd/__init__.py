"""mustafar_amd -- MI355X-native (gfx950) implementation of Mustafar's sparse-attention decode path.

Public surface mirrors the reference's operator boundary:
  mustafar_amd.mustafar_package   <-> kernel/kernel_wrapper (mustafar_key_formulation / mustafar_value_formulation)
  mustafar_amd.compression        <-> kernel/compression.py (convert_key_batched / convert_value_batched)
  mustafar_amd.hook               <-> decode/prefill branches of models/llama_mustafar_kernel.py
The compute lives in mustafar_amd/csrc/*.hip behind the C ABI of include/mustafar_hip.h.
"""
__version__ = "0.1.0"
