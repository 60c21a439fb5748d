mkdir -p gpurun_out; export HSA_ENABLE_IPC_MODE_LEGACY=0 TMPDIR=/tmp
export AFFNET_PARITY_REPORT=$PWD/gpurun_out/parity_report_onepass.json
timeout 900 python -m pytest tests/test_gpu_onepass.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -rA -k "onepass or local_norm or dense_affnet or nms2d or pyramid_variants" > gpurun_out/pytest_onepass.log 2>&1; echo "pytest exit: $?"; tail -n 12 gpurun_out/pytest_onepass.log | cut -c1-250
grep -n "^E " gpurun_out/pytest_onepass.log | cut -c1-600 | head -20
