set -u
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r03_pytest_gpu.log
echo "pytest rc=$?" >> gpurun_out/r03_pytest_gpu.log
python bench.py > gpurun_out/r03_bench_plain.json 2> gpurun_out/r03_bench_plain.err
echo "bench rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-traffic > gpurun_out/r03_bench_torchrun_world1.json 2> gpurun_out/r03_bench_torchrun_world1.err
echo "torchrun rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --scaling strong --no-traffic > gpurun_out/r03_bench_torchrun_world1_strong.json 2> gpurun_out/r03_bench_torchrun_world1_strong.err
echo "torchrun strong rc=$?"
tests/cpp/build/sharded_stats 1000000000 1 weak 20 > gpurun_out/r03_sharded_stats_weak.json 2> gpurun_out/r03_sharded_stats.err
echo "cpp weak rc=$?"
tests/cpp/build/sharded_stats 1000000000 1 strong 20 > gpurun_out/r03_sharded_stats_strong.json 2>> gpurun_out/r03_sharded_stats.err
echo "cpp strong rc=$?"
tail -5 gpurun_out/r03_pytest_gpu.log
