# Regenerates the round's evidence under gpurun_out/refresh (one gpurun call); tools/collect_profiles.py copies the results to profiles/<round>_*.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
MT_BENCH_DETAIL=$O/default_detail.json timeout -k 10 300 python bench.py > $O/default_line.json 2> $O/default.err
MT_BENCH_DETAIL=$O/driver_detail.json timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_line.json 2> $O/driver.err
MT_BENCH_DETAIL=$O/1stream_detail.json timeout -k 10 200 python bench.py --streams 1 --cosched 1 --no-cpu-baseline --no-sections > $O/1stream_line.json 2>/dev/null
MT_BENCH_DETAIL=$O/1forward_detail.json timeout -k 10 200 python bench.py --streams 1 --cosched 4 --no-cpu-baseline --no-sections > $O/1forward_line.json 2>/dev/null
timeout -k 10 200 python bench.py --mode train --batch 16 --steps 10 --warmup 2 > $O/train_line.json 2>/dev/null
timeout -k 10 200 python bench.py --mode train --model cnn_rnn_large --batch 16 --steps 8 --warmup 2 > $O/train_large_line.json 2>/dev/null
timeout -k 10 200 python bench.py --model cnn_rnn_large --batch 16 > $O/large_line.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
export MT_BENCH_DETAIL=$O/scratch_detail.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_default -- python3 $R/bench.py --no-cpu-baseline > $O/p_default.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_1stream -- python3 $R/bench.py --streams 1 --cosched 1 --no-cpu-baseline --no-sections > $O/p_1stream.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_1forward -- python3 $R/bench.py --streams 1 --cosched 4 --no-cpu-baseline --no-sections > $O/p_1forward.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_train -- python3 $R/bench.py --mode train --batch 16 --steps 10 --warmup 2 > $O/p_train.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_train_large -- python3 $R/bench.py --mode train --model cnn_rnn_large --batch 16 --steps 8 --warmup 2 > $O/p_train_large.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_large -- python3 $R/bench.py --model cnn_rnn_large --batch 16 > $O/p_large.log 2>&1
cd $R
python3 tools/trace_exclusive.py $O/p_train_large 6 > $O/train_large_exclusive_time.txt 2>&1 || true
python3 tools/trace_by_shape.py $O/p_train_large 10 > $O/train_large_by_shape.txt 2>&1 || true
python3 tools/trace_exclusive.py $O/p_train 8 > $O/train_exclusive_time.txt 2>&1 || true
rm -f $O/*/*/*kernel_trace.csv $O/*/*/*domain_stats.csv $O/scratch_detail.json
ls $O
