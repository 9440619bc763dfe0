set -x
mkdir -p gpurun_out/r03
python tools/ntt_ab_lib.py tools/ab/libgstark_hip_r02.so 24 > gpurun_out/r03/ntt_ab_lib.txt 2>&1
tail -35 gpurun_out/r03/ntt_ab_lib.txt
GSTARK_NTT_TWIDDLE_LOG=24 python tools/ntt_time.py > gpurun_out/r03/ntt_time_twlog24.txt 2>&1; tail -12 gpurun_out/r03/ntt_time_twlog24.txt
bash tools/ntt_prof.sh 24 > gpurun_out/r03/ntt_prof_2p24.txt 2>&1; cat gpurun_out/r03/ntt_prof_2p24.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "ntt or lazy or lde" 2>&1 | tail -5
