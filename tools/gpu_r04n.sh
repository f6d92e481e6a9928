O=gpurun_out/r04n; mkdir -p $O
timeout 900 bash tools/profile_r04.sh r04n_bc7 bc7
timeout 1200 bash tools/profile_r04.sh r04n_others others
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --no-cpu-baseline --cfg5-images 16 > $O/bench_2ranks.json 2> $O/bench_2ranks.err; echo "2rank rc=$?"
