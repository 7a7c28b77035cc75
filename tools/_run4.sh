set -e -o pipefail
mkdir -p gpurun_out/r4d
python -m pytest tests/test_gpu_engines.py tests/test_gpu_extents.py tests/test_gpu_mask.py "tests/test_gpu_benchshape.py::test_fused_arena_eager_and_graph_at_bench_shape" -x -q > gpurun_out/r4d/tests.log 2>&1 || { tail -30 gpurun_out/r4d/tests.log; exit 1; }
tail -2 gpurun_out/r4d/tests.log
V=$PWD/mustafar_amd/lib/variants
for rep in 1 2; do
python tools/quick.py --cfg c3 --set dot2 mfma 2>> gpurun_out/r4d/q.err | grep cfg | sed 's/^/base    /' | tee -a gpurun_out/r4d/q.txt
MUSTAFAR_HIP_LIB=$V/libmustafar_hip_nozfill.so python tools/quick.py --cfg c3 --set dot2 mfma 2>> gpurun_out/r4d/q.err | grep cfg | sed 's/^/nozfill /' | tee -a gpurun_out/r4d/q.txt
MUSTAFAR_HIP_LIB=$V/libmustafar_hip_noprio.so python tools/quick.py --cfg c3 --set dot2 mfma 2>> gpurun_out/r4d/q.err | grep cfg | sed 's/^/noprio  /' | tee -a gpurun_out/r4d/q.txt
done
python tools/quick.py --cfg c4 c5 --set dot2 mfma 2>> gpurun_out/r4d/q.err | grep cfg | sed 's/^/base    /' | tee -a gpurun_out/r4d/q.txt
MUSTAFAR_HIP_LIB=$V/libmustafar_hip_nozfill.so python tools/quick.py --cfg c4 c5 --set dot2 mfma 2>> gpurun_out/r4d/q.err | grep cfg | sed 's/^/nozfill /' | tee -a gpurun_out/r4d/q.txt
