#!/bin/bash
# Build the `mustafar_package` PyTorch extension (host C++ only) against libmustafar_hip.so.
# Output: mustafar_amd/dropin/mustafar_package<EXT_SUFFIX>  (put mustafar_amd/dropin on PYTHONPATH for a drop-in).
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../dropin"
mkdir -p "$OUT"
TORCH=$(python3 -c "import torch,os; print(os.path.dirname(torch.__file__))")
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
SUF=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
g++ -O2 -std=c++17 -fPIC -shared -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 -D_GLIBCXX_USE_CXX11_ABI=$(python3 -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))") \
    -DTORCH_EXTENSION_NAME=mustafar_package -DTORCH_API_INCLUDE_EXTENSION_H \
    -I"$TORCH/include" -I"$TORCH/include/torch/csrc/api/include" -I/opt/rocm/include -I"$PYINC" \
    "$HERE/torch_ext.cpp" -o "$OUT/mustafar_package$SUF" \
    -L"$TORCH/lib" -ltorch -ltorch_cpu -lc10 -lc10_hip -ltorch_hip -ltorch_python \
    -L"$HERE/../lib" -lmustafar_hip -Wl,-rpath,'$ORIGIN/../lib' -Wl,-rpath,"$TORCH/lib" -Wno-deprecated-declarations
