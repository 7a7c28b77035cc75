"""Packaging of the drop-in boundary, the counterpart of the reference's kernel/kernel_wrapper/setup.py:23-41:

    pip package   mustafar_batched_spmv_package          (setup.py:24)
    import name   mustafar_package                        (setup.py:28; imported at models/llama_mustafar_kernel.py:14)

The reference compiles pybind.cpp + mustafar_wrapper.cu with nvcc and links ../build/SpMM_API.o (the CUDA kernels).  Here the
extension is HOST code only (mustafar_amd/csrc/torch_ext.cpp) on top of the C ABI of include/mustafar_hip.h; the kernels live
in libmustafar_hip.so, built for gfx950 by mustafar_amd/csrc/build.sh (hipcc cross-compiles without a GPU) and linked here.

    python setup.py build_ext --build-lib mustafar_amd/dropin     # in-tree build (what __graft_entry__.build() runs)
    pip install -e .                                              # mustafar_amd + the compiled mustafar_package module
"""
import os
import subprocess

import torch
from setuptools import find_packages, setup
from torch.utils.cpp_extension import BuildExtension, CppExtension

ROOT = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(ROOT, "mustafar_amd", "lib")
TORCH_LIB = os.path.join(os.path.dirname(torch.__file__), "lib")


class BuildWithKernels(BuildExtension):
    """libmustafar_hip.so first (the extension links it), then the host extension."""

    def run(self):
        subprocess.check_call(["bash", os.path.join(ROOT, "mustafar_amd", "csrc", "build.sh")])
        super().run()


setup(
    name="mustafar_batched_spmv_package",           # pip package name (reference setup.py:24)
    version="0.2.0",
    description="MI355X (gfx950) drop-in for the Mustafar batched SpMV extension",
    packages=find_packages(include=["mustafar_amd", "mustafar_amd.*"]),
    package_data={"mustafar_amd": ["lib/libmustafar_hip.so"]},
    ext_modules=[
        CppExtension(
            name="mustafar_package",                 # import module name (reference setup.py:28)
            sources=[os.path.join("mustafar_amd", "csrc", "torch_ext.cpp")],
            include_dirs=[os.path.join(ROOT, "include"), "/opt/rocm/include"],
            define_macros=[("__HIP_PLATFORM_AMD__", "1"), ("USE_ROCM", "1")],
            library_dirs=[LIB_DIR],
            libraries=["mustafar_hip", "c10_hip", "torch_hip"],
            extra_compile_args=["-O2", "-g0", "-std=c++17", "-Wno-deprecated-declarations"],
            # in-tree (mustafar_amd/dropin/mustafar_package*.so -> ../lib) and installed (site-packages -> mustafar_amd/lib)
            extra_link_args=["-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,$ORIGIN/mustafar_amd/lib", f"-Wl,-rpath,{LIB_DIR}",
                             f"-Wl,-rpath,{TORCH_LIB}"],
        )
    ],
    cmdclass={"build_ext": BuildWithKernels.with_options(use_ninja=False)},
    install_requires=["torch"],
)
