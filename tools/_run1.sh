set -e
mkdir -p gpurun_out/r4a
python -m pytest tests/test_gpu_extents.py tests/test_gpu_engines.py tests/test_gpu_benchshape.py -x -q > gpurun_out/r4a/tests.log 2>&1 || { tail -30 gpurun_out/r4a/tests.log; exit 1; }
tail -3 gpurun_out/r4a/tests.log
python tools/quick.py --cfg c3 --set dot2 dot2:dyn=0 dot2:dwl=1 mfma mfma:dyn=0 valu valu:dyn=0 dot2 dot2:dyn=0 2> gpurun_out/r4a/quick_c3.err | grep cfg > gpurun_out/r4a/quick_c3.txt; cat gpurun_out/r4a/quick_c3.txt
python tools/quick.py --cfg c4 c5 --set dot2 dot2:dyn=0 mfma mfma:dyn=0 2> gpurun_out/r4a/quick_c45.err | grep cfg > gpurun_out/r4a/quick_c45.txt; cat gpurun_out/r4a/quick_c45.txt
