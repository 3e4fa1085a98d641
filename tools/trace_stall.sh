# kernel traces of the Large training step (a) inside the default bench run (behind the other sections) and (b) in a fresh process
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stall
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/t_default -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/default.log 2>&1
tail -c 600 $O/default.log
python3 $R/tools/trace_queues.py $O/t_default 4 > $O/default_queues.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t_alone -- python3 $R/bench.py --mode train --model cnn_rnn_large --batch 16 --steps 6 --warmup 2 > $O/alone.log 2>&1
python3 $R/tools/trace_queues.py $O/t_alone 4 > $O/alone_queues.txt 2>&1 || true
rm -rf $O/t_default $O/t_alone
cat $O/default_queues.txt $O/alone_queues.txt
