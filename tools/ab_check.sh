# A/B of a kernel change on ONE box: build the previous commit as a variant first
#   git stash && bash tools/build_asm_variant.sh before "" && git stash pop
# then on the GPU box: the parity subset (forms, parity, ASR against the oracle) and bench.py's K1 time of the product against
# linearham_amd/lib_exp/before on configs[2], a 145-pattern family and the ragged-read family (tools/var_check.sh).
set -e
python -m pytest tests/test_gpu_forms.py tests/test_gpu_parity.py tests/test_gpu_asr.py -q -x > gpurun_out/p3_tests.log 2>&1 || { tail -30 gpurun_out/p3_tests.log; exit 1; }
tail -2 gpurun_out/p3_tests.log
bash tools/var_check.sh
bash tools/var_check.sh --brlen-mean 0.003
bash tools/var_check.sh --preset config2_ragged
