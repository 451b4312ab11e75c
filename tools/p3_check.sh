set -e
python -m pytest tests/test_gpu_forms.py tests/test_gpu_parity.py tests/test_gpu_asr.py -q -x > gpurun_out/p3_tests.log 2>&1 || { tail -30 gpurun_out/p3_tests.log; exit 1; }
tail -2 gpurun_out/p3_tests.log
bash tools/var_check.sh
bash tools/var_check.sh --brlen-mean 0.003
bash tools/var_check.sh --preset config2_ragged
