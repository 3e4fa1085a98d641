# PMC passes for profiles/ (each counter set in its own run, kernel trace only beside it)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# usage: refresh_pmc.sh [headline|sections|all]   (two gpurun calls keep each under the call's time limit)
PART=${1:-all}
# ONE forward in flight -- the configuration the bench line's `stages` / `roofline` are measured in:
#   CMD4: the forward shape of the default timed region (4 batches of 32 per forward);  CMD1: one batch of 32 per forward
CMD4="bench.py --steps 8 --warmup 4 --cosched 4 --streams 1 --no-cpu-baseline --no-sections"
CMD1="bench.py --steps 3 --warmup 1 --cosched 1 --streams 1 --no-cpu-baseline --no-sections"
if [ $PART = headline ] || [ $PART = all ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/$CMD4 > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/$CMD4 > $O/write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/mfma -- python3 $R/$CMD4 > $O/mfma.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch1 -- python3 $R/$CMD1 > $O/fetch1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write1 -- python3 $R/$CMD1 > $O/write1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/mfma1 -- python3 $R/$CMD1 > $O/mfma1.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/fetch $O/write $O/pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 $CMD4" 4 > $O/traffic.txt
python3 tools/pmc_mfma_summary.py $O/mfma $O/pmc_mfma_util.json "rocprofv3 --kernel-trace --pmc MfmaUtil -- python3 $CMD4" > $O/mfma.txt
python3 tools/pmc_summary.py $O/fetch1 $O/write1 $O/pmc_traffic_b32.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 $CMD1" 1 >> $O/traffic.txt
python3 tools/pmc_mfma_summary.py $O/mfma1 $O/pmc_mfma_util_b32.json "rocprofv3 --kernel-trace --pmc MfmaUtil -- python3 $CMD1" >> $O/mfma.txt
rm -rf $O/fetch $O/write $O/mfma $O/fetch1 $O/write1 $O/mfma1
cat $O/mfma.txt
cd /tmp
fi
if [ $PART = sections ] || [ $PART = all ]; then
# the Large forward (BASELINE configs[2]) and the training step (configs[3]): the kernels the sections' rooflines name
CMDL="bench.py --model cnn_rnn_large --batch 16 --steps 3 --warmup 1 --streams 1"
CMDT="bench.py --mode train --batch 16 --steps 3 --warmup 1"
CMDU="bench.py --mode train --model cnn_rnn_large --batch 16 --steps 3 --warmup 1"
for tag in L T U; do
  if [ $tag = L ]; then CMD="$CMDL"; elif [ $tag = T ]; then CMD="$CMDT"; else CMD="$CMDU"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch$tag -- python3 $R/$CMD > $O/fetch$tag.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write$tag -- python3 $R/$CMD > $O/write$tag.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/mfma$tag -- python3 $R/$CMD > $O/mfma$tag.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $O/fetchL $O/writeL $O/pmc_traffic_large.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 $CMDL" 1 > $O/traffic_lt.txt
python3 tools/pmc_mfma_summary.py $O/mfmaL $O/pmc_mfma_util_large.json "rocprofv3 --kernel-trace --pmc MfmaUtil -- python3 $CMDL" > $O/mfma_lt.txt
python3 tools/pmc_summary.py $O/fetchT $O/writeT $O/pmc_traffic_train.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 $CMDT" 1 >> $O/traffic_lt.txt
python3 tools/pmc_mfma_summary.py $O/mfmaT $O/pmc_mfma_util_train.json "rocprofv3 --kernel-trace --pmc MfmaUtil -- python3 $CMDT" >> $O/mfma_lt.txt
python3 tools/pmc_summary.py $O/fetchU $O/writeU $O/pmc_traffic_train_large.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 $CMDU" 1 >> $O/traffic_lt.txt
python3 tools/pmc_mfma_summary.py $O/mfmaU $O/pmc_mfma_util_train_large.json "rocprofv3 --kernel-trace --pmc MfmaUtil -- python3 $CMDU" >> $O/mfma_lt.txt
rm -rf $O/fetchL $O/writeL $O/mfmaL $O/fetchT $O/writeT $O/mfmaT $O/fetchU $O/writeU $O/mfmaU
cat $O/mfma_lt.txt
fi
