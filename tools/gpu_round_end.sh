# Round-end evidence run on the GPU box (through gpurun): rocprofv3 profiles of the BC7 pipeline and of the other workloads (tools/profile.sh), the
# default bench line, and the N > 1 bench path with two ranks sharing the GPU over gloo. Outputs under gpurun_out/; copy what is to be judged into profiles/.
O=gpurun_out/round_end; mkdir -p $O
timeout 900 bash tools/profile.sh ${TAG:-round_end}_bc7 bc7
timeout 1200 bash tools/profile.sh ${TAG:-round_end}_others others
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python tools/perf_guard.py $O/bench.json > $O/perf_guard.txt 2>&1; echo "perf_guard rc=$?"; tail -25 $O/perf_guard.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --no-cpu-baseline --cfg5-images 16 > $O/bench_2ranks.out 2> $O/bench_2ranks.err; echo "2rank rc=$?"
grep '^{"metric"' $O/bench_2ranks.out > $O/bench_2ranks.json      # (gloo prints its connection chatter on stdout before the line)
