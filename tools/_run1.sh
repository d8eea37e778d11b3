set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r03_t5.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/r03_t5.log
timeout -k 10 300 python tools/fuzz_gpu.py 3000 3031 > gpurun_out/r03_fuzz_auto.json 2> gpurun_out/r03_fuzz1.err; echo "fuzz rc=$?"; head -c 500 gpurun_out/r03_fuzz_auto.json
CBA_LM_RESIDENT=0 timeout -k 10 300 python tools/fuzz_gpu.py 3000 3032 > gpurun_out/r03_fuzz_host_driven.json 2> gpurun_out/r03_fuzz2.err; echo "fuzz rc=$?"; head -c 500 gpurun_out/r03_fuzz_host_driven.json
timeout -k 10 300 python tools/fuzz_small.py > gpurun_out/r03_fuzz_small.json 2> gpurun_out/r03_fuzz3.err; echo "fuzz small rc=$?"; head -c 600 gpurun_out/r03_fuzz_small.json
