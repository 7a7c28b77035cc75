"""`kernel` package shim: lets `import kernel.compression as compression` (models/llama_mustafar_kernel.py:19) resolve
to the MI355X implementation when mustafar_amd/dropin is on PYTHONPATH."""
