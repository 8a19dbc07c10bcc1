set -x
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_fullsize.py tests/test_boundary.py -m gpu -q -s -x 2>&1 | tail -80 > gpurun_out/r02/fullsize_tests_v0.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r02/bench_det512_v0.json 2> gpurun_out/r02/bench_det512_v0.err
python bench.py --workload seg1024tiled --steps 5 --warmup 2 > gpurun_out/r02/bench_seg1024_v0.json 2> gpurun_out/r02/bench_seg1024_v0.err
python bench.py --workload det512s50 --steps 3 --warmup 1 > gpurun_out/r02/bench_s50_v0.json 2> gpurun_out/r02/bench_s50_v0.err
EDTR_BENCH_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/r02/bench_dist1_v0.json 2> gpurun_out/r02/bench_dist1_v0.err
tail -3 gpurun_out/r02/*.json gpurun_out/r02/fullsize_tests_v0.log
