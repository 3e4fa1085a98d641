# PMC passes for profiles/ (each counter set in its own run, kernel trace only beside it)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
# (one batch of 32 per forward, one forward in flight: the kernels the bench line's `stages` / `roofline` are measured on)
CMD="bench.py --steps 3 --warmup 1 --cosched 1 --streams 1 --no-cpu-baseline --no-sections"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/$CMD > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/$CMD > $O/write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc MfmaUtil --output-format csv -d $O/mfma -- python3 $R/$CMD > $O/mfma.log 2>&1
cd $R
python3 tools/pmc_summary.py $O/fetch $O/write $O/pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 $CMD" > $O/traffic.txt
python3 tools/pmc_mfma_summary.py $O/mfma $O/pmc_mfma_util.json "rocprofv3 --kernel-trace --pmc MfmaUtil -- python3 $CMD" > $O/mfma.txt
rm -rf $O/fetch $O/write $O/mfma
cat $O/mfma.txt
