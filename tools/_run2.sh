set -e -o pipefail
mkdir -p gpurun_out/r4b
MUSTAFAR_HIP_LIB=$PWD/mustafar_amd/lib/variants/libmustafar_hip_dynprobe.so python tools/quick.py --cfg c3 --set dot2:dwl=1 dot2:dyn=0 dot2:dwl=1 2> gpurun_out/r4b/q2.err | grep cfg | tee gpurun_out/r4b/q2.txt
MUSTAFAR_HIP_LIB=$PWD/mustafar_amd/lib/variants/libmustafar_hip_tick1.so python tools/quick.py --cfg c3 --set dot2 dot2:dwl=1 dot2:dyn=0 2> gpurun_out/r4b/q3.err | grep cfg | tee gpurun_out/r4b/q3.txt
