set -e -o pipefail
mkdir -p gpurun_out/r4c
python tools/quick.py --cfg c3 --set dot2 dot2:prio=1 dot2:prio=2 dot2:prio=3 dot2 mfma mfma:prio=1 mfma:prio=2 2> gpurun_out/r4c/q1.err | grep cfg | tee gpurun_out/r4c/q1.txt
python tools/quick.py --cfg c4 c5 --set dot2 dot2:prio=1 dot2:prio=2 mfma mfma:prio=1 2> gpurun_out/r4c/q2.err | grep cfg | tee gpurun_out/r4c/q2.txt
