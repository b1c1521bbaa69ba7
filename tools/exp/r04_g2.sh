python -m pytest tests/test_gpu_overlap.py tests/test_gpu_gather.py tests/test_gpu_texture.py tests/test_gpu_features.py -x -q -m gpu --durations=15 > gpurun_out/r04_t2.txt 2>&1
tail -30 gpurun_out/r04_t2.txt
bash tools/exp/r04_g3.sh
