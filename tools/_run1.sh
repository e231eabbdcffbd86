python -m pytest tests/test_agg_gpu.py -q -m gpu -k "batch_form" -x 2>&1 | tail -3 > gpurun_out/f3_6.txt
export DSMIL_NATIVE_LIB=libdsmil_hip_expt.so
DSMIL_BATCH_FORM=2 python tools/f2_ablate.py masks 0 >> gpurun_out/f3_6.txt 2>&1
DSMIL_BATCH_FORM=1 python tools/f2_ablate.py masks 0 >> gpurun_out/f3_6.txt 2>&1
DSMIL_BATCH_FORM=2 python tools/f2_ablate.py masks 0 >> gpurun_out/f3_6.txt 2>&1
DSMIL_F3_DBG=1 DSMIL_EXPT=64 python tools/f3_stamps.py 2>&1 | grep -v amdgpu >> gpurun_out/f3_6.txt
cat gpurun_out/f3_6.txt
